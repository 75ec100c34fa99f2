#!/bin/bash
# round 6, job e: prefill kernel with the shared odd fragment (+ tiled operands): kernel tests, bit-identity, A/B
R="${GRAFT_REPO_ROOT:-.}"; O=$R/gpurun_out/r6_e; mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_gpu_kernels.py -m gpu -x -q -k "gemm_dma or gemm_ring or plane_output or post_norm or gemm_bf16x2" > $O/pytest_gemm.txt 2>&1
tail -5 $O/pytest_gemm.txt
timeout 600 python tools/tiled_check.py 798 7 > $O/r6_tiled_check.txt 2>&1
tail -8 $O/r6_tiled_check.txt | cut -c1-400
