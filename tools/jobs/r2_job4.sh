#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 600 python tools/ts_gemm_sweep.py 16 128 > gpurun_out/r2_ts_gemm_sweep.txt 2>&1
cat gpurun_out/r2_ts_gemm_sweep.txt
timeout 600 python -m pytest tests/test_gpu_server.py -x -q > gpurun_out/r2_server_tests.log 2>&1
tail -15 gpurun_out/r2_server_tests.log
