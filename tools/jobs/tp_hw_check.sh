#!/bin/bash
# Hardware checklist for the FIRST run on a multi-GPU node (nothing below has ever crossed a real xGMI link: gpurun boxes have
# one GPU; DESIGN.md section 6 lists what was verified on one device).  Run from the repo root on an N-GPU MI355X node:
#     bash tools/jobs/tp_hw_check.sh [N]        (N defaults to the number of visible GPUs)
# Every step writes into gpurun_out/tp_hw/ and prints one PASS / FAIL line; stop at the first FAIL and read its log.
N="${1:-$(python -c 'import torch; print(torch.cuda.device_count())')}"
R="${GRAFT_REPO_ROOT:-.}"; cd "$R"; O=gpurun_out/tp_hw; mkdir -p "$O"
export HSA_ENABLE_IPC_MODE_LEGACY=0
say() { if [ "$1" -eq 0 ]; then echo "PASS  $2"; else echo "FAIL  $2   (see $3)"; fi; }
echo "== $N GPUs =="
# 1. RCCL itself over this node's links (torch.distributed, backend nccl): the fallback path of every prefill-sized all-reduce
timeout 300 python bench.py --gpus "$N" --launch-check > "$O/1_launch.json" 2> "$O/1_launch.err"; say $? "bench.py --gpus $N launches $N ranks, RCCL all-reduce" "$O/1_launch.err"
# 2. the exchange kernels' own tests (both ranks in one process on device 0: a sanity check of this box before the links are involved;
#    the cross-DEVICE form of the same kernels is step 4, through IPC-mapped buffers)
timeout 600 python -m pytest tests/test_gpu_tp_p2p.py -x -q -m gpu > "$O/2_p2p_tests.log" 2>&1; say $? "exchange kernels, single device (tests/test_gpu_tp_p2p.py)" "$O/2_p2p_tests.log"
# 3. latency of one exchange per peer pair (software cost was 4.9 us on one device; the xGMI hop comes on top)
timeout 300 python tools/tp_exchange_bench.py > "$O/3_exchange_latency.txt" 2>&1; say $? "exchange latency (tools/tp_exchange_bench.py)" "$O/3_exchange_latency.txt"
# 4. the IPC-mapped cross-process exchange + whole-step hipGraph, TP = 2 / 4 / N: tokens must equal the committed full-depth oracle
#    run (parity_checked true), tp_exchange must say "p2p one-shot kernels", status word clean
for W in 2 4 "$N"; do
  [ "$W" -le "$N" ] || continue
  timeout 900 python bench.py --gpus "$W" --no-cpu-baseline --steps 32 --warmup 8 > "$O/4_bench_tp$W.json" 2> "$O/4_bench_tp$W.err"
  rc=$?
  if [ $rc -eq 0 ]; then python - "$O/4_bench_tp$W.json" <<'PY'
import json, sys
d = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
ok = d["parity_checked"] and d["n_gpus"] == d["config"]["rccl_world_size"] and "p2p" in (d["config"]["tp_exchange"] or "")
print("     n_gpus", d["n_gpus"], "tokens/s", round(d["value"], 1), "ttft_ms", round(d["ttft_ms_p50"], 1), "exchange:", d["config"]["tp_exchange"], "parity_checked", d["parity_checked"])
print("     release form of the prefill-sized sums (P2PExchange.first_contact, round 6):", d["config"]["tp_release"])
sys.exit(0 if ok else 1)
PY
  rc=$?; fi
  say $rc "bench.py --gpus $W (TP=$W decode graph + p2p exchange, oracle tokens)" "$O/4_bench_tp$W.err"
done
# 4b. the same with the release form FORCED each way (round 6: step 4's form was decided by the first-contact test - "light (validated ...)"
#     or "fence (...)", printed above; CHATTS_TP_BULK_FENCE=1 / 0 overrides it: compare ttft_ms / parity_checked.  Forcing 0 on a node whose
#     first-contact test chose the fence is expected to FAIL parity - that is the test's finding, not a bug of this script)
for W in 2 "$N"; do
  [ "$W" -le "$N" ] || continue
  CHATTS_TP_BULK_FENCE=1 timeout 900 python bench.py --gpus "$W" --no-cpu-baseline --steps 16 --warmup 4 > "$O/4b_bench_tp${W}_fence.json" 2> "$O/4b_bench_tp${W}_fence.err"
  say $? "bench.py --gpus $W with TP_BULK_FENCE=1 (compare ttft_ms / parity_checked with step 4)" "$O/4b_bench_tp${W}_fence.err"
done
# 5. the plain-Python call shape of the reference (demo/demo_vllm.py:30): LLM(tensor_parallel_size=2) spawns its follower
timeout 300 python tools/llm_tp_spawn_check.py > "$O/5_llm_spawn.json" 2> "$O/5_llm_spawn.err"; say $? "LLM(tensor_parallel_size=2) from one process" "$O/5_llm_spawn.err"
# 6. the TP server under its own launcher (leader announces engine iterations, followers replay)
timeout 600 python tools/tp2_server_check.py > "$O/6_server.json" 2> "$O/6_server.err"; say $? "TP=2 OpenAI server == TP=1 engine" "$O/6_server.err"
