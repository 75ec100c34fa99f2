#!/bin/bash
# bench.py's own TP path (self-launch, one process per rank, IPC-mapped exchange) at FULL depth with all ranks on the one GPU, round-6 code:
# (1) as is - the exchange knows its ranks share a device: light release, no test sums; (2) CHATTS_TP_ASSUME_CROSS_DEVICE=1 - the
# cross-device branch of P2PExchange.first_contact: 64 test sums under both forms, light granted, recorded in config.tp_release;
# (3) the same with CHATTS_TP_INJECT_RELEASE_MISMATCH=1 - the fence stays; (4) bench.py's own second line (tokens against the oracle run)
# with an injected verdict.  Tokens must equal the committed full-depth oracle run every time.  Correctness evidence only (ranks time-slice one device).
R="${GRAFT_REPO_ROOT:-.}"; O=$R/gpurun_out/r6_tp; mkdir -p $O; cd $R
export HSA_ENABLE_IPC_MODE_LEGACY=0 CHATTS_FORCE_DEVICE=0 CHATTS_DIST_BACKEND=gloo CHATTS_TP_FUSE_BLOCKS=48 CHATTS_TP_BULK_BLOCKS=16 CHATTS_TP_AR_BLOCKS=16
show() { python - "$1" <<'PY'
import json, sys
try:
    d = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
    c = d["config"]
    print("   n_gpus", d["n_gpus"], "tok/s", round(d["value"], 1), "ttft", round(d["ttft_ms_p50"], 1), "parity_checked", d["parity_checked"], "graph", c["decode_graph"],
          "status", c.get("tp_status"), "exchange:", (c["tp_exchange"] or "")[:30], "\n   release:", c.get("tp_release"))
except Exception as e:
    print("FAILED", e); print(open(sys.argv[1].replace(".json", ".err")).read()[-1500:])
PY
}
run() { name=$1; shift; env "$@" timeout 900 python bench.py --gpus $W --steps 16 --warmup 4 --no-cpu-baseline --ttft-runs 2 > $O/$name.json 2> $O/$name.err; echo "$name rc=$?"; show $O/$name.json; }
W=2; run r6_tp2_self_launch_single_device X=0
W=8; run r6_tp8_self_launch_single_device X=0
W=2; run r6_tp2_first_contact_cross_device CHATTS_TP_ASSUME_CROSS_DEVICE=1
W=8; run r6_tp8_first_contact_cross_device CHATTS_TP_ASSUME_CROSS_DEVICE=1
W=2; run r6_tp2_first_contact_injected_mismatch CHATTS_TP_ASSUME_CROSS_DEVICE=1 CHATTS_TP_INJECT_RELEASE_MISMATCH=1
W=2; run r6_tp2_bench_injected_token_mismatch CHATTS_TP_ASSUME_CROSS_DEVICE=1 CHATTS_BENCH_INJECT_RELEASE_MISMATCH=1
grep -n "repeating with" $O/r6_tp2_bench_injected_token_mismatch.err | head -3
