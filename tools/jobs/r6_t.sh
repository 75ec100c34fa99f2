#!/bin/bash
# round 6, job t: gemm_f16q_kernel on tiled weights (1 KB pieces) - bit-identity + A/B against the row-major form and the bf16x2 kernel
R="${GRAFT_REPO_ROOT:-.}"; O=$R/gpurun_out/r6_t; mkdir -p $O
cd $R
timeout 600 python tools/f16q_check.py 798 5 > $O/f16q_check_tiled.txt 2>&1
tail -12 $O/f16q_check_tiled.txt | cut -c1-330
