#!/bin/bash
R="${GRAFT_REPO_ROOT:-.}"; cd "$R"; mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_kernels.py -x -q --timeout 300 -k "gemm or linear or plane or ts_encoder" 2>&1 | tail -12 > gpurun_out/r3_gemm32_pytest.log; tail -5 gpurun_out/r3_gemm32_pytest.log
timeout 300 python tools/gemm_dma_sweep.py 798 > gpurun_out/r3_gemm_dma32_sweep.txt 2>&1; cat gpurun_out/r3_gemm_dma32_sweep.txt | tail -40
