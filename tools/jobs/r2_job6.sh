#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out
export TMPDIR=/tmp
true
true
timeout 400 python -m pytest tests/test_gpu_e2e.py tests/test_gptq.py -x -q --timeout 150 > gpurun_out/r2_job6_e2e.log 2>&1
tail -25 gpurun_out/r2_job6_e2e.log
