#!/bin/bash
# round-4 evidence refresh after the planes attention (prefill bits changed): full-depth parity of the headline and 8B lines, the bench
# lines, kernel trace, the config-5 FETCH_SIZE pass (attention.hip is in its digest), per-rank TP steps.  Summaries are copied into
# profiles/ by hand afterwards.
R="${GRAFT_REPO_ROOT:-.}"; O=$R/gpurun_out; mkdir -p $O
export PMC_TRAFFIC_OUT=$O/pmc_traffic.json HSA_ENABLE_IPC_MODE_LEGACY=0
cp $R/profiles/pmc_traffic.json $O/pmc_traffic.json
cd $R
timeout 700 python tools/parity_full_depth.py --model chatts-14b --out $O/r4_parity_14b_8x256_bf16_b1_full.json > $O/r4_parity_14b.log 2>&1; echo "parity 14b rc=$?"; tail -2 $O/r4_parity_14b.log | cut -c1-300
timeout 500 python tools/parity_full_depth.py --model chatts-8b --series 1 --length 256 --out $O/r4_parity_8b_1x256_bf16_b1_full.json > $O/r4_parity_8b.log 2>&1; echo "parity 8b rc=$?"; tail -2 $O/r4_parity_8b.log | cut -c1-300
cd /tmp && export TMPDIR=/tmp
( cd $R && timeout 300 python bench.py --steps 32 --warmup 8 --no-cpu-baseline > $O/r4_bench_n1_no_cpu_baseline.json 2> $O/r4_bench_n1.err ); echo "bench rc=$?"
rm -rf /tmp/kt; C1="python bench.py --steps 8 --warmup 2 --no-cpu-baseline --ttft-runs 3"
timeout 300 rocprofv3 --kernel-trace -d /tmp/kt -o p -- python $R/bench.py --steps 8 --warmup 2 --no-cpu-baseline --ttft-runs 3 > /tmp/kt.log 2>&1
(echo "## rocprofv3 --kernel-trace -- $C1   (MI355X, round 4, final code)"; python $R/tools/prof_db.py $(find /tmp/kt -name "*.db" | head -1)) > $O/r4_bench_kernel_trace.txt
rm -rf /tmp/fs5; C5="python bench.py --batch 16 --weights fp8 --series 8 --length 1024 --steps 4 --warmup 2 --no-cpu-baseline"
timeout 500 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d /tmp/fs5 -o p -- python $R/bench.py --batch 16 --weights fp8 --series 8 --length 1024 --steps 4 --warmup 2 --no-cpu-baseline > /tmp/fs5.log 2>&1
db=$(find /tmp/fs5 -name "*.db" | head -1)
(echo "## rocprofv3 --kernel-trace --pmc FETCH_SIZE -- $C5   (MI355X, round 4, final code)"; python $R/tools/prof_db.py $db | grep -v fill_hash | head -60) > $O/r4_cfg5_pmc_fetch_size.txt
( cd $R && python tools/pmc_traffic.py batched $db "rocprofv3 --kernel-trace --pmc FETCH_SIZE -- $C5" profiles/r4_cfg5_pmc_fetch_size.txt )
cd $R
timeout 300 python bench.py --batch 16 --weights fp8 --series 8 --length 1024 --steps 24 --warmup 6 --no-cpu-baseline > $O/r4_bench_cfg5_fp8_8x1024_b16.json 2> $O/r4_bench_cfg5.err
timeout 300 python bench.py --series 30 --lengths mixed --steps 32 --warmup 8 --no-cpu-baseline > $O/r4_bench_cfg4_30xmixed.json 2> $O/r4_bench_cfg4.err
timeout 200 python bench.py --model chatts-8b --series 1 --length 256 --steps 32 --warmup 8 --no-cpu-baseline > $O/r4_bench_8b_cfg2.json 2> $O/r4_bench_8b.err
timeout 400 python tools/tp_shard_step.py --worlds 1,2,4,8 --out $O/r4_tp_shard_step.json > /dev/null 2> $O/r4_tp_shard_step.err
python - <<'PY'
import json, glob, os
O = os.path.join(os.environ.get("GRAFT_REPO_ROOT", "."), "gpurun_out")
for f in ("r4_bench_n1_no_cpu_baseline", "r4_bench_cfg5_fp8_8x1024_b16", "r4_bench_cfg4_30xmixed", "r4_bench_8b_cfg2"):
    try:
        r = json.loads(open(f"{O}/{f}.json").read().strip().splitlines()[-1])
        print(f, {k: r.get(k) for k in ("value", "ms_per_step", "ttft_ms_p50", "parity_checked")}, (r.get("roofline") or {}).get("traffic"))
    except Exception as e:
        print(f, "ERR", e)
try:
    r = json.load(open(f"{O}/r4_tp_shard_step.json"))
    print({w: (round(v["prefill_ms"], 2), round(v["decode_ms_per_step"], 3)) for w, v in r["worlds"].items()})
except Exception as e:
    print("tp ERR", e)
PY
