#!/bin/bash
# the optional precision="bf16" speed mode: kernel test + same-process comparison with the parity-grade default
mkdir -p gpurun_out
timeout 300 python -m pytest tests/test_gpu_kernels.py -x -q --timeout 200 -k "gemm_dma_single_pass or gemm_dma_parity or gemm_dma_equals" 2>&1 | tail -3
timeout 300 python tools/speed_mode_check.py chatts-14b 2> gpurun_out/r2_speed_mode.err | tail -1 > gpurun_out/r2_speed_mode_14b.json
cat gpurun_out/r2_speed_mode_14b.json; tail -3 gpurun_out/r2_speed_mode.err
