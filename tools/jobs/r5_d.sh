#!/bin/bash
# round 5, call D: the matrix pipe's power-bound rate by operand data; interleaved A/B of the prefill projections
set -x
mkdir -p gpurun_out/r5_d
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 300 tools/probes/mfma_power_probe > gpurun_out/r5_d/mfma_power_probe.txt 2>&1
cat gpurun_out/r5_d/mfma_power_probe.txt
timeout 600 python tools/gemm_ab.py 798 7 ring_T6=T:6 > gpurun_out/r5_d/gemm_ab.txt 2>&1
cat gpurun_out/r5_d/gemm_ab.txt
