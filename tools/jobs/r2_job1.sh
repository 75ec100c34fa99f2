#!/bin/bash
# round-2 GPU job 1: real-size parity tests, full-depth oracle runs, MALL probe, balanced GEMV geometry sweep
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out
export TMPDIR=/tmp
(free -g; nproc) > gpurun_out/r2_box.txt 2>&1
timeout 900 python -m pytest tests/test_gpu_parity_real_size.py -x -q > gpurun_out/r2_parity_tests.log 2>&1
tail -5 gpurun_out/r2_parity_tests.log
timeout 600 python -m pytest tests/test_gpu_kernels.py -x -q -k "gemv" > gpurun_out/r2_gemv_tests.log 2>&1
tail -3 gpurun_out/r2_gemv_tests.log
timeout 900 python tools/parity_full_depth.py --model chatts-14b --out gpurun_out/r2_parity_14b_full.json > gpurun_out/r2_parity_14b_full.log 2>&1
tail -c 600 gpurun_out/r2_parity_14b_full.log
timeout 600 python tools/parity_full_depth.py --model chatts-8b --series 1 --out gpurun_out/r2_parity_8b_full.json > gpurun_out/r2_parity_8b_full.log 2>&1
tail -c 400 gpurun_out/r2_parity_8b_full.log
timeout 300 python tools/mall_probe.py > gpurun_out/r2_mall_probe.txt 2>&1
cat gpurun_out/r2_mall_probe.txt
timeout 600 python tools/gemv_sweep.py quick > gpurun_out/r2_gemv_sweep.txt 2>&1
cat gpurun_out/r2_gemv_sweep.txt
