#!/bin/bash
# round 5, call C: ring kernel - who waits for whom (probe build), kernel trace of the bench with it, remaining shapes of the check,
# and the one-launch decode attention (kernel test + bench A/B)
set -x
mkdir -p gpurun_out/r5_c
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
CHATTS_AMD_LIB=chatts_amd/lib/variants/libchatts_amd_probe.so timeout 300 python tools/ring_probe.py 798 > gpurun_out/r5_c/ring_probe.txt 2>&1
cat gpurun_out/r5_c/ring_probe.txt
timeout 600 python -m pytest tests/test_gpu_kernels.py -q -m gpu -k "attention_decode" > gpurun_out/r5_c/pytest_attn.txt 2>&1
tail -15 gpurun_out/r5_c/pytest_attn.txt
timeout 300 rocprofv3 --kernel-trace -d /tmp/kt -o p -- python bench.py --steps 8 --warmup 2 --no-cpu-baseline --ttft-runs 3 > gpurun_out/r5_c/bench_fold.txt 2>&1
python tools/prof_db.py $(find /tmp/kt -name "*.db" | head -1) > gpurun_out/r5_c/kt_ring_fold.txt
head -40 gpurun_out/r5_c/kt_ring_fold.txt
tail -1 gpurun_out/r5_c/bench_fold.txt | cut -c1-400
CHATTS_ATTN_FOLD=0 timeout 300 python bench.py --steps 16 --warmup 2 --no-cpu-baseline --ttft-runs 3 > gpurun_out/r5_c/bench_nofold.txt 2>&1
tail -1 gpurun_out/r5_c/bench_nofold.txt | cut -c1-400
timeout 300 python bench.py --steps 16 --warmup 2 --no-cpu-baseline --ttft-runs 3 > gpurun_out/r5_c/bench_fold2.txt 2>&1
tail -1 gpurun_out/r5_c/bench_fold2.txt | cut -c1-400
timeout 900 python tools/gemm_ring_check.py > gpurun_out/r5_c/ring_check.txt 2>&1
tail -60 gpurun_out/r5_c/ring_check.txt
