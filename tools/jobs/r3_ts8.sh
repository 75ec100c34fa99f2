#!/bin/bash
R="${GRAFT_REPO_ROOT:-.}"; cd $R; mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_kernels.py -q -x -m gpu -k "multiblock or ts_encode or stream" 2>&1 | tail -3
timeout 300 python tools/ts_gemm_sweep.py 128 64 32 2>&1 | grep -v amdgpu.ids | grep -v "SK=1 \|SK=2 \|SK=3 \|SK=16\|SK=20\|SK=12" | tee gpurun_out/r3_ts_gemm_sweep.txt
