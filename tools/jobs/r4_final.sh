#!/bin/bash
# round-4 final evidence on the final code: the bench line as the driver runs it, kernel traces, FETCH_SIZE passes with code digests
# (-> gpurun_out/pmc_traffic.json, copied to profiles/ afterwards), full-depth parity of the headline and 8B lines, the other workloads'
# bench lines, the per-rank TP steps.  Summaries are copied into profiles/ by hand afterwards.
R="${GRAFT_REPO_ROOT:-.}"; O=$R/gpurun_out; mkdir -p $O
export PMC_TRAFFIC_OUT=$O/pmc_traffic.json HSA_ENABLE_IPC_MODE_LEGACY=0
cd /tmp && export TMPDIR=/tmp
( cd $R && timeout 600 python bench.py --steps 32 --warmup 8 > $O/r4_bench_n1.json 2> $O/r4_bench_n1.err ); echo "bench rc=$?"
rm -rf /tmp/kt; C1="python bench.py --steps 8 --warmup 2 --no-cpu-baseline --ttft-runs 3"
timeout 300 rocprofv3 --kernel-trace -d /tmp/kt -o p -- python $R/bench.py --steps 8 --warmup 2 --no-cpu-baseline --ttft-runs 3 > /tmp/kt.log 2>&1
(echo "## rocprofv3 --kernel-trace -- $C1   (MI355X, round 4, final code)"; python $R/tools/prof_db.py $(find /tmp/kt -name "*.db" | head -1)) > $O/r4_bench_kernel_trace.txt
rm -rf /tmp/fs; C2="python bench.py --steps 4 --warmup 2 --no-cpu-baseline --ttft-runs 1"
timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d /tmp/fs -o p -- python $R/bench.py --steps 4 --warmup 2 --no-cpu-baseline --ttft-runs 1 > /tmp/fs.log 2>&1
db=$(find /tmp/fs -name "*.db" | head -1)
(echo "## rocprofv3 --kernel-trace --pmc FETCH_SIZE -- $C2   (MI355X, round 4, final code)"; python $R/tools/prof_db.py $db) > $O/r4_bench_pmc_fetch_size.txt
python $R/tools/pmc_traffic.py headline $db "rocprofv3 --kernel-trace --pmc FETCH_SIZE -- $C2" profiles/r4_bench_pmc_fetch_size.txt
rm -rf /tmp/fst
TS_CALLS=20 timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d /tmp/fst -o p -- python $R/tools/pmc_traffic.py ts-run > /tmp/fst.log 2>&1
db=$(find /tmp/fst -name "*.db" | head -1)
(echo "## rocprofv3 --kernel-trace --pmc FETCH_SIZE -- python tools/pmc_traffic.py ts-run   (TS encoder alone, 8 x 256, 21 calls, MI355X, round 4)"; python $R/tools/prof_db.py $db) > $O/r4_ts_encoder_pmc_fetch_size.txt
TS_CALLS=20 python $R/tools/pmc_traffic.py ts $db "rocprofv3 --kernel-trace --pmc FETCH_SIZE -- python tools/pmc_traffic.py ts-run" profiles/r4_ts_encoder_pmc_fetch_size.txt
rm -rf /tmp/fs5; C5="python bench.py --batch 16 --weights fp8 --series 8 --length 1024 --steps 4 --warmup 2 --no-cpu-baseline"
timeout 500 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d /tmp/fs5 -o p -- python $R/bench.py --batch 16 --weights fp8 --series 8 --length 1024 --steps 4 --warmup 2 --no-cpu-baseline > /tmp/fs5.log 2>&1
db=$(find /tmp/fs5 -name "*.db" | head -1)
(echo "## rocprofv3 --kernel-trace --pmc FETCH_SIZE -- $C5   (MI355X, round 4, final code)"; python $R/tools/prof_db.py $db | grep -v fill_hash | head -60) > $O/r4_cfg5_pmc_fetch_size.txt
python $R/tools/pmc_traffic.py batched $db "rocprofv3 --kernel-trace --pmc FETCH_SIZE -- $C5" profiles/r4_cfg5_pmc_fetch_size.txt
rm -rf /tmp/kt5
timeout 400 rocprofv3 --kernel-trace -d /tmp/kt5 -o p -- python $R/bench.py --batch 16 --weights fp8 --series 8 --length 1024 --steps 8 --warmup 2 --no-cpu-baseline > /tmp/kt5.log 2>&1
(echo "## rocprofv3 --kernel-trace -- python bench.py --batch 16 --weights fp8 --series 8 --length 1024 --steps 8 --warmup 2 --no-cpu-baseline  (MI355X, round 4, final code)"; python $R/tools/prof_db.py $(find /tmp/kt5 -name "*.db" | head -1)) > $O/r4_cfg5_kernel_trace.txt
cd $R
timeout 900 python tools/parity_full_depth.py --model chatts-14b --out $O/r4_parity_14b_8x256_bf16_b1_full.json > $O/r4_parity_14b.log 2>&1; echo "parity 14b rc=$?"; tail -2 $O/r4_parity_14b.log | cut -c1-400
timeout 900 python tools/parity_full_depth.py --model chatts-8b --series 1 --length 256 --out $O/r4_parity_8b_1x256_bf16_b1_full.json > $O/r4_parity_8b.log 2>&1; echo "parity 8b rc=$?"; tail -2 $O/r4_parity_8b.log | cut -c1-400
timeout 400 python bench.py --batch 16 --weights fp8 --series 8 --length 1024 --steps 24 --warmup 6 --no-cpu-baseline > $O/r4_bench_cfg5_fp8_8x1024_b16.json 2> $O/r4_bench_cfg5.err
timeout 400 python bench.py --batch 16 --weights fp8 --precision fp8 --series 8 --length 1024 --steps 24 --warmup 6 --no-cpu-baseline > $O/r4_bench_cfg5_fp8_speed_mode.json 2> $O/r4_bench_cfg5_speed.err
timeout 400 python bench.py --series 30 --lengths mixed --steps 32 --warmup 8 --no-cpu-baseline > $O/r4_bench_cfg4_30xmixed.json 2> $O/r4_bench_cfg4.err
timeout 300 python bench.py --model chatts-8b --series 1 --length 256 --steps 32 --warmup 8 --no-cpu-baseline > $O/r4_bench_8b_cfg2.json 2> $O/r4_bench_8b.err
timeout 300 python bench.py --weights fp8 --steps 32 --warmup 8 --no-cpu-baseline > $O/r4_bench_fp8_weights.json 2> $O/r4_bench_fp8.err
timeout 300 python bench.py --weights int4 --steps 32 --warmup 8 --no-cpu-baseline > $O/r4_bench_int4_weights.json 2> $O/r4_bench_int4.err
timeout 600 python tools/tp_shard_step.py --worlds 1,2,4,8 --out $O/r4_tp_shard_step.json > /dev/null 2> $O/r4_tp_shard_step.err
timeout 400 python tools/tp_shard_step.py --worlds 8 --batch 16 --weights fp8 --prefill-runs 1 --out $O/r4_tp8_shard_step_cfg5.json > /dev/null 2> $O/r4_tp8_shard_step_cfg5.err
cd /tmp; rm -rf /tmp/kt8
timeout 600 rocprofv3 --kernel-trace -d /tmp/kt8 -o p -- python $R/tools/tp_shard_step.py --worlds 8 --steps 8 --warmup 2 --prefill-runs 2 --out $O/r4_tp8_traced.json > /tmp/kt8.log 2>&1
(echo "## rocprofv3 --kernel-trace -- python tools/tp_shard_step.py --worlds 8 --steps 8 --warmup 2 --prefill-runs 2   (ONE rank of TP=8, loop-back exchange, MI355X, round 4, final code)"; python $R/tools/prof_db.py $(find /tmp/kt8 -name "*.db" | head -1)) > $O/r4_tp8_shard_kernel_trace.txt
cd $R
python - <<PY
import json
for f in ("r4_bench_n1","r4_bench_cfg5_fp8_8x1024_b16","r4_bench_cfg5_fp8_speed_mode","r4_bench_cfg4_30xmixed","r4_bench_8b_cfg2","r4_bench_fp8_weights","r4_bench_int4_weights"):
    try:
        d=json.loads(open("gpurun_out/%s.json"%f).read().strip().splitlines()[-1])
        print(f, round(d["value"],1), round(d["ms_per_step"],3), "ttft", d.get("ttft_ms_p50"), "parity", d.get("parity_checked"), "frac", round(d["roofline"]["frac"],3), "traffic", d["roofline"].get("traffic"), (d.get("cpu_baseline") or {}).get("value"))
    except Exception as e:
        print(f, "FAILED", e)
for f in ("r4_parity_14b_8x256_bf16_b1_full", "r4_parity_8b_1x256_bf16_b1_full"):
    try:
        d=json.load(open("gpurun_out/%s.json"%f)); print(f, {k: d[k] for k in d if "err" in k or k in ("tokens_match",)})
    except Exception as e:
        print(f, "FAILED", e)
print(open("gpurun_out/r4_tp_shard_step.err").read()[-1500:])
PY
