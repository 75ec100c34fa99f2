#!/bin/bash
# bench.py's own TP path (self-launch, one process per rank, IPC-mapped exchange) at FULL depth with all ranks on the one GPU, round-5 code:
# tokens must equal the committed full-depth oracle run (parity_checked); then the same with an injected "light release gave different
# tokens" verdict on the first attempt: every rank must switch to the system-scope fence, repeat the stage and say so (config.tp_release).
# Correctness evidence only - the ranks time-slice one device.
R="${GRAFT_REPO_ROOT:-.}"; O=$R/gpurun_out/r5_tp; mkdir -p $O; cd $R
export HSA_ENABLE_IPC_MODE_LEGACY=0 CHATTS_FORCE_DEVICE=0 CHATTS_DIST_BACKEND=gloo CHATTS_TP_FUSE_BLOCKS=48 CHATTS_TP_BULK_BLOCKS=16 CHATTS_TP_AR_BLOCKS=16
show() { python - "$1" <<'PY'
import json, sys
try:
    d = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
    c = d["config"]
    print("n_gpus", d["n_gpus"], "tok/s", round(d["value"], 1), "ttft", round(d["ttft_ms_p50"], 1), "parity_checked", d["parity_checked"], "graph", c["decode_graph"],
          "release:", c.get("tp_release"), "status", c.get("tp_status"), "exchange:", (c["tp_exchange"] or "")[:40])
except Exception as e:
    print("FAILED", e); print(open(sys.argv[1].replace(".json", ".err")).read()[-1500:])
PY
}
for W in 2 8; do
  timeout 900 python bench.py --gpus $W --steps 16 --warmup 4 --no-cpu-baseline --ttft-runs 2 > $O/r5_tp${W}_self_launch_single_device.json 2> $O/r5_tp${W}_self_launch_single_device.err; echo "bench --gpus $W rc=$?"
  show $O/r5_tp${W}_self_launch_single_device.json
done
CHATTS_BENCH_INJECT_RELEASE_MISMATCH=1 timeout 900 python bench.py --gpus 2 --steps 16 --warmup 4 --no-cpu-baseline --ttft-runs 2 > $O/r5_tp2_self_launch_injected_release_mismatch.json 2> $O/r5_tp2_self_launch_injected_release_mismatch.err; echo "bench --gpus 2 (injected) rc=$?"
show $O/r5_tp2_self_launch_injected_release_mismatch.json
grep -n "repeating with" $O/r5_tp2_self_launch_injected_release_mismatch.err | head -3
