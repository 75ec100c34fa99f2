#!/bin/bash
# after the last attention change (1-D heaviest-first XCD-aware grid, swizzled LDS): tests that run prefill attention, bench + trace,
# config-5 FETCH_SIZE pass (attention.hip / attn_decode.h are in its digest)
R="${GRAFT_REPO_ROOT:-.}"; O=$R/gpurun_out; mkdir -p $O
export PMC_TRAFFIC_OUT=$O/pmc_traffic.json HSA_ENABLE_IPC_MODE_LEGACY=0
cp $R/profiles/pmc_traffic.json $O/pmc_traffic.json
cd $R
timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_parity_real_size.py tests/test_gpu_tp_shards.py tests/test_gpu_tp_p2p.py -x -q -m gpu -k "attention or attn or parity or shards or prefill" 2>&1 | tail -3 | tee $O/r4_final3_tests.txt
cd /tmp && export TMPDIR=/tmp
( cd $R && timeout 300 python bench.py --steps 32 --warmup 8 --no-cpu-baseline > $O/r4_bench_n1_no_cpu_baseline.json 2> $O/r4_bench_n1.err ); echo "bench rc=$?"
rm -rf /tmp/kt; C1="python bench.py --steps 8 --warmup 2 --no-cpu-baseline --ttft-runs 3"
timeout 300 rocprofv3 --kernel-trace -d /tmp/kt -o p -- python $R/bench.py --steps 8 --warmup 2 --no-cpu-baseline --ttft-runs 3 > /tmp/kt.log 2>&1
(echo "## rocprofv3 --kernel-trace -- $C1   (MI355X, round 4, final code)"; python $R/tools/prof_db.py $(find /tmp/kt -name "*.db" | head -1)) > $O/r4_bench_kernel_trace.txt
rm -rf /tmp/fs5; C5="python bench.py --batch 16 --weights fp8 --series 8 --length 1024 --steps 4 --warmup 2 --no-cpu-baseline"
timeout 500 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d /tmp/fs5 -o p -- python $R/bench.py --batch 16 --weights fp8 --series 8 --length 1024 --steps 4 --warmup 2 --no-cpu-baseline > /tmp/fs5.log 2>&1
db=$(find /tmp/fs5 -name "*.db" | head -1)
(echo "## rocprofv3 --kernel-trace --pmc FETCH_SIZE -- $C5   (MI355X, round 4, final code)"; python $R/tools/prof_db.py $db | grep -v fill_hash | head -60) > $O/r4_cfg5_pmc_fetch_size.txt
( cd $R && python tools/pmc_traffic.py batched $db "rocprofv3 --kernel-trace --pmc FETCH_SIZE -- $C5" profiles/r4_cfg5_pmc_fetch_size.txt ) | cut -c1-200
cd $R
timeout 300 python bench.py --series 30 --lengths mixed --steps 32 --warmup 8 --no-cpu-baseline > $O/r4_bench_cfg4_30xmixed.json 2> $O/r4_bench_cfg4.err
timeout 400 python tools/tp_shard_step.py --worlds 1,8 --steps 16 --out $O/r4_tp_shard_step_w1_w8.json > /dev/null 2> $O/r4_tp_shard_step.err
python - <<'PY'
import json, os
O = os.path.join(os.environ.get("GRAFT_REPO_ROOT", "."), "gpurun_out")
for f in ("r4_bench_n1_no_cpu_baseline", "r4_bench_cfg4_30xmixed"):
    try:
        r = json.loads(open(f"{O}/{f}.json").read().strip().splitlines()[-1])
        print(f, {k: r.get(k) for k in ("value", "ms_per_step", "ttft_ms_p50", "parity_checked")}, (r.get("roofline") or {}).get("traffic"))
    except Exception as e:
        print(f, "ERR", e)
try:
    r = json.load(open(f"{O}/r4_tp_shard_step_w1_w8.json"))
    print({w: (round(v["prefill_ms"], 2), round(v["decode_ms_per_step"], 3)) for w, v in r["worlds"].items()})
except Exception as e:
    print("tp ERR", e)
PY
grep -i "attn_prefill\|kv_planes" $O/r4_bench_kernel_trace.txt | cut -c1-150
