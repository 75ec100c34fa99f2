#!/bin/bash
# round 5, call H: (workgroups, threads) sweep of the bulk all-reduce on a loop-back TP = 8 / 2 rank; the TS encoder on the prefill
# kernel instead of the weight-streaming one (A/B); roctx stage summary
set -x
O=gpurun_out/r5_h; mkdir -p $O
cd /tmp && export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0 && cd $GRAFT_REPO_ROOT
timeout 300 python tools/tp_bulk_sweep.py 8 798 > $O/tp_bulk_sweep_w8.txt 2>&1; cat $O/tp_bulk_sweep_w8.txt
timeout 300 python tools/tp_bulk_sweep.py 2 798 > $O/tp_bulk_sweep_w2.txt 2>&1; cat $O/tp_bulk_sweep_w2.txt
for mb in default 0; do
  if [ $mb = default ]; then unset CHATTS_GEMM_STREAM_MB; else export CHATTS_GEMM_STREAM_MB=$mb; fi
  timeout 300 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --ttft-runs 1 > $O/bench_ts_$mb.txt 2>&1
  python - $O/bench_ts_$mb.txt <<'PY'
import json, sys
r = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(sys.argv[1], {k: r["ts_encoder_roofline"].get(k) for k in ("avg_us", "frac", "gbs", "achieved")}, "ttft", r["ttft_ms_p50"])
PY
done
unset CHATTS_GEMM_STREAM_MB
rm -rf /tmp/ktm
timeout 300 rocprofv3 --kernel-trace --marker-trace -d /tmp/ktm -o p -- python bench.py --steps 4 --warmup 2 --no-cpu-baseline --ttft-runs 2 > $O/bench_marker.log 2>&1
(echo "## rocprofv3 --kernel-trace --marker-trace -- python bench.py --steps 4 --warmup 2 --no-cpu-baseline --ttft-runs 2   (MI355X, round 5)"; python tools/prof_db.py $(find /tmp/ktm -name "*.db" | head -1)) > $O/r5_bench_marker_trace.txt
tail -16 $O/r5_bench_marker_trace.txt
