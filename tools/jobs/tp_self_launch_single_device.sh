#!/bin/bash
# bench.py's own TP path (self-launch, one process per rank, IPC-mapped exchange) at FULL depth with all ranks on the one GPU:
# tokens must equal the committed full-depth oracle run (parity_checked).  Correctness evidence only - the ranks time-slice one device.
R="${GRAFT_REPO_ROOT:-.}"; O=$R/gpurun_out; mkdir -p $O; cd $R
export HSA_ENABLE_IPC_MODE_LEGACY=0 CHATTS_FORCE_DEVICE=0 CHATTS_DIST_BACKEND=gloo CHATTS_TP_FUSE_BLOCKS=48 CHATTS_TP_BULK_BLOCKS=16 CHATTS_TP_AR_BLOCKS=16
for W in 2 8; do
  timeout 900 python bench.py --gpus $W --steps 16 --warmup 4 --no-cpu-baseline --ttft-runs 2 > $O/r4_tp${W}_self_launch_single_device.json 2> $O/r4_tp${W}_self_launch_single_device.err; echo "bench --gpus $W rc=$?"
  python - <<PY
import json
try:
    d=json.loads([l for l in open("$O/r4_tp${W}_self_launch_single_device.json") if l.startswith("{")][-1])
    print("n_gpus", d["n_gpus"], "tok/s", round(d["value"],1), "ttft", round(d["ttft_ms_p50"],1), "parity_checked", d["parity_checked"], "graph", d["config"]["decode_graph"], "exchange:", d["config"]["tp_exchange"][:60])
except Exception as e:
    print("FAILED", e); print(open("$O/r4_tp${W}_self_launch_single_device.err").read()[-1500:])
PY
done
