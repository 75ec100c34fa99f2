#!/bin/bash
# round 5, call B: first run of gemm_ring_kernel - values, bits, time; then the GEMM tests, real-size parity, bench
set -x
mkdir -p gpurun_out/r5_b
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 900 python tools/gemm_ring_check.py > gpurun_out/r5_b/ring_check.txt 2>&1
tail -120 gpurun_out/r5_b/ring_check.txt
timeout 900 python -m pytest tests/test_gpu_kernels.py -q -m gpu -k "linear or gemm or plane or ts_enc" --maxfail=40 -x -q > gpurun_out/r5_b/pytest_gemm.txt 2>&1
tail -30 gpurun_out/r5_b/pytest_gemm.txt
timeout 900 python -m pytest tests/test_gpu_parity_real_size.py -q -m gpu --maxfail=10 > gpurun_out/r5_b/pytest_parity.txt 2>&1
tail -15 gpurun_out/r5_b/pytest_parity.txt
timeout 600 python bench.py --steps 8 --warmup 2 --no-cpu-baseline > gpurun_out/r5_b/bench_ring.txt 2>&1
tail -3 gpurun_out/r5_b/bench_ring.txt
CHATTS_GEMM_RING=0 timeout 600 python bench.py --steps 8 --warmup 2 --no-cpu-baseline > gpurun_out/r5_b/bench_old.txt 2>&1
tail -3 gpurun_out/r5_b/bench_old.txt
