#!/bin/bash
# round 6, job d: the prefill kernel on tiled operands - bit-identity + A/B
R="${GRAFT_REPO_ROOT:-.}"; O=$R/gpurun_out/r6_d; mkdir -p $O
cd $R
timeout 600 python tools/tiled_check.py 798 7 > $O/r6_tiled_check.txt 2>&1
tail -20 $O/r6_tiled_check.txt | cut -c1-400
