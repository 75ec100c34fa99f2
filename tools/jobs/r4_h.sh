#!/bin/bash
# round 4, call H: fp8 GEMM kernel tests + where the fp8 prefill's time goes (kernel trace at 4 layers)
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out; O=gpurun_out; R=$PWD
timeout 900 python -m pytest tests/test_gpu_kernels.py -m gpu -q -k "fp8_speed or quantize_rows_fp8" > $O/r4h_fp8_tests.log 2>&1; echo "fp8 kernel tests rc=$?"; tail -12 $O/r4h_fp8_tests.log | cut -c1-300
cd /tmp && export TMPDIR=/tmp; rm -rf /tmp/kt9
timeout 600 rocprofv3 --kernel-trace -d /tmp/kt9 -o p -- python $R/tools/fp8_speed_mode.py 4 > /tmp/kt9.log 2>&1; echo "rocprof rc=$?"
db=$(find /tmp/kt9 -name "*.db" | head -1)
(echo "## rocprofv3 --kernel-trace -- python tools/fp8_speed_mode.py 4   (4 layers, both modes in one process, MI355X, round 4)"; python $R/tools/prof_db.py $db) > $R/$O/r4h_fp8_kernel_trace.txt
cd $R; head -40 $O/r4h_fp8_kernel_trace.txt | cut -c1-230
