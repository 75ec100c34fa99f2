#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 300 python -m pytest tests/test_gpu_server.py -x -q --timeout 100 > gpurun_out/r2_job6_server.log 2>&1
tail -25 gpurun_out/r2_job6_server.log
timeout 400 python -m pytest tests/test_gpu_e2e.py tests/test_gptq.py -x -q --timeout 150 > gpurun_out/r2_job6_e2e.log 2>&1
tail -25 gpurun_out/r2_job6_e2e.log
