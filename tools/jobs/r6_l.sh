#!/bin/bash
# round 6, job l: the opt-in f16q prefill mode at full depth (ChatTS-14B, 8 x 256): parity against the CPU float32 oracle, bench line;
# and the default mode's full-depth parity re-run on the round-6 code (tiled weights, down_proj (6, 2))
R="${GRAFT_REPO_ROOT:-.}"; O=$R/gpurun_out/r6_l; mkdir -p $O
cd $R
timeout 1500 python tools/parity_full_depth.py --precision f16q --out $O/r6_parity_14b_8x256_bf16_b1_f16q_full.json > $O/parity_f16q.log 2>&1
tail -4 $O/parity_f16q.log | cut -c1-300
timeout 1500 python tools/parity_full_depth.py --out $O/r6_parity_14b_8x256_bf16_b1_full.json > $O/parity_default.log 2>&1
tail -4 $O/parity_default.log | cut -c1-300
