#!/bin/bash
# round 4, call G: fp8 speed mode (kernel tests, measurement), multi-process TP tests with capped grids, attention refactor regression
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out; O=gpurun_out; R=$PWD
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 900 python -m pytest tests/test_gpu_kernels.py -m gpu -q -x -k "fp8_speed or quantize_rows_fp8" > $O/r4g_fp8_tests.log 2>&1; echo "fp8 kernel tests rc=$?"; tail -15 $O/r4g_fp8_tests.log | cut -c1-300
timeout 900 python tools/fp8_speed_mode.py > $O/r4g_fp8_mode.log 2>&1; echo "fp8 speed mode rc=$?"; tail -3 $O/r4g_fp8_mode.log | cut -c1-1800
timeout 1200 python -m pytest tests/test_gpu_tp_multiprocess.py -m gpu -q -x > $O/r4g_mp_tests.log 2>&1; echo "multiprocess tests rc=$?"; tail -5 $O/r4g_mp_tests.log | cut -c1-600
timeout 900 python -m pytest tests/test_gpu_e2e.py -m gpu -q -x > $O/r4g_e2e.log 2>&1; echo "e2e rc=$?"; tail -3 $O/r4g_e2e.log | cut -c1-300
