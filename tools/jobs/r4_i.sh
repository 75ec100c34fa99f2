#!/bin/bash
# round 4, call I: fp8 GEMM v2 (128/256 x 256 tiles, 8 waves, loads two K-steps ahead): kernel tests, speed-mode measurement, tile A/B
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out; O=gpurun_out; R=$PWD
timeout 900 python -m pytest tests/test_gpu_kernels.py -m gpu -q -k "fp8_speed or quantize_rows_fp8" > $O/r4i_fp8_tests.log 2>&1; echo "fp8 kernel tests rc=$?"; tail -4 $O/r4i_fp8_tests.log | cut -c1-300
timeout 900 python tools/fp8_speed_mode.py > $O/r4i_fp8_mode.log 2>&1; echo "fp8 speed mode rc=$?"; tail -1 $O/r4i_fp8_mode.log | cut -c1-1200

CHATTS_FP8_BM=128 timeout 900 python tools/fp8_speed_mode.py 12 > $O/r4i_fp8_mode_bm128.log 2>&1; echo "bm128 (12 layers)"; tail -1 $O/r4i_fp8_mode_bm128.log | cut -c1-700
cd /tmp && export TMPDIR=/tmp; rm -rf /tmp/kt9
timeout 600 rocprofv3 --kernel-trace -d /tmp/kt9 -o p -- python $R/tools/fp8_speed_mode.py 4 > /tmp/kt9.log 2>&1; echo "rocprof rc=$?"
db=$(find /tmp/kt9 -name "*.db" | head -1)
(echo "## rocprofv3 --kernel-trace -- python tools/fp8_speed_mode.py 4   (4 layers, both modes in one process, MI355X, round 4)"; python $R/tools/prof_db.py $db) > $R/$O/r4i_fp8_kernel_trace.txt
cd $R; grep "gemm_fp8\|quantize" $O/r4i_fp8_kernel_trace.txt | cut -c1-160
