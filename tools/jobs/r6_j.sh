#!/bin/bash
# round 6, job j: 8-bit weight decode GEMV geometry (rows x chunks in flight per lane, waves per workgroup) - bench lines, fp8 weights batch 1
R="${GRAFT_REPO_ROOT:-.}"; O=$R/gpurun_out/r6_j; mkdir -p $O
cd $R
i=0
for cfg in "" "CHATTS_GEMV8_ROWS=4 CHATTS_GEMV8_UNR=2" "CHATTS_GEMV8_ROWS=4 CHATTS_GEMV8_UNR=1" "CHATTS_GEMV8_ROWS=2 CHATTS_GEMV8_UNR=4" \
           "CHATTS_GEMV8_ROWS=4 CHATTS_GEMV8_UNR=2 CHATTS_GEMV8_NW=8" "CHATTS_GEMV8_ROWS=4 CHATTS_GEMV8_UNR=2 CHATTS_GEMV8_NW=16" "CHATTS_GEMV8_NW=8" "CHATTS_GEMV8_NW=16"; do
  i=$((i+1))
  env $cfg timeout 400 python bench.py --weights fp8 --steps 40 --warmup 5 --no-cpu-baseline --ttft-runs 1 > $O/b$i.json 2> $O/b$i.err
  python - "$cfg" $O/b$i.json <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[2]).read().strip().splitlines()[-1])
    print(f"{sys.argv[1] or 'default':70s} tok/s {d['value']:7.2f}  ms/step {d['ms_per_step']:.3f}  parity {d['parity_checked']}  dominant {d['roofline']['avg_us']:.1f} us frac {d['roofline']['frac']:.3f}")
except Exception as e:
    print(sys.argv[1], "failed", e)
PY
done | tee $O/sweep.txt
