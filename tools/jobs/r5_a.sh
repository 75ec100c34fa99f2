#!/bin/bash
# round 5, call A: where the prefill GEMM's time goes (timeline probe build + K sweep)
set -x
mkdir -p gpurun_out/r5_a
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
CHATTS_AMD_LIB=chatts_amd/lib/variants/libchatts_amd_probe.so timeout 300 python tools/gemm_probe.py 798 > gpurun_out/r5_a/gemm_probe.txt 2>&1
timeout 400 python tools/gemm_ksweep.py > gpurun_out/r5_a/gemm_ksweep.txt 2>&1
tail -50 gpurun_out/r5_a/gemm_probe.txt
tail -40 gpurun_out/r5_a/gemm_ksweep.txt
