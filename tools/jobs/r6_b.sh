#!/bin/bash
# round 6, job b: first contact of gemm_f16q_kernel - correctness against the float64 reference + A/B against gemm_ring_kernel
R="${GRAFT_REPO_ROOT:-.}"; O=$R/gpurun_out/r6_b; mkdir -p $O
cd $R
timeout 600 python tools/f16q_check.py 798 5 > $O/f16q_check.txt 2>&1
tail -20 $O/f16q_check.txt | cut -c1-400
