#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out
timeout 300 python -m pytest tests/test_gpu_kernels.py -x -q --timeout 100 -k "bf16x3 or attention_parity" > gpurun_out/r2_job19.log 2>&1
tail -12 gpurun_out/r2_job19.log
timeout 100 python tools/attn_prefill_time.py 2>&1 | tail -5
