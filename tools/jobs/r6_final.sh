#!/bin/bash
# round-6 final evidence on the final code: the bench line as the driver runs it, kernel traces, FETCH_SIZE passes with code digests
# (-> gpurun_out/r6_final/pmc_traffic.json, copied to profiles/ afterwards), full-depth parity of every workload that has a bench line,
# the other workloads' bench lines, the per-rank TP steps.  Summaries are copied into profiles/ by hand afterwards.
R="${GRAFT_REPO_ROOT:-.}"; O=$R/gpurun_out/r6_final; mkdir -p $O
export PMC_TRAFFIC_OUT=$O/pmc_traffic.json HSA_ENABLE_IPC_MODE_LEGACY=0
cp $R/profiles/pmc_traffic.json $O/pmc_traffic.json
cd /tmp && export TMPDIR=/tmp
( cd $R && timeout 600 python bench.py --steps 32 --warmup 8 > $O/r6_bench_n1.json 2> $O/r6_bench_n1.err ); echo "bench rc=$?"
rm -rf /tmp/kt; C1="python bench.py --steps 8 --warmup 2 --no-cpu-baseline --ttft-runs 3"
timeout 300 rocprofv3 --kernel-trace -d /tmp/kt -o p -- python $R/bench.py --steps 8 --warmup 2 --no-cpu-baseline --ttft-runs 3 > /tmp/kt.log 2>&1
(echo "## rocprofv3 --kernel-trace -- $C1   (MI355X, round 6, final code)"; python $R/tools/prof_db.py $(find /tmp/kt -name "*.db" | head -1)) > $O/r6_bench_kernel_trace.txt
rm -rf /tmp/fs; C2="python bench.py --steps 4 --warmup 2 --no-cpu-baseline --ttft-runs 1"
timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d /tmp/fs -o p -- python $R/bench.py --steps 4 --warmup 2 --no-cpu-baseline --ttft-runs 1 > /tmp/fs.log 2>&1
db=$(find /tmp/fs -name "*.db" | head -1)
(echo "## rocprofv3 --kernel-trace --pmc FETCH_SIZE -- $C2   (MI355X, round 6, final code)"; python $R/tools/prof_db.py $db) > $O/r6_bench_pmc_fetch_size.txt
( cd $R && python tools/pmc_traffic.py headline $db "rocprofv3 --kernel-trace --pmc FETCH_SIZE -- $C2" profiles/r6_bench_pmc_fetch_size.txt )
rm -rf /tmp/fst
( cd $R && TS_CALLS=20 timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d /tmp/fst -o p -- python tools/pmc_traffic.py ts-run > /tmp/fst.log 2>&1 )
db=$(find /tmp/fst -name "*.db" | head -1)
(echo "## rocprofv3 --kernel-trace --pmc FETCH_SIZE -- python tools/pmc_traffic.py ts-run   (TS encoder alone, 8 x 256, 21 calls, MI355X, round 6)"; python $R/tools/prof_db.py $db) > $O/r6_ts_encoder_pmc_fetch_size.txt
( cd $R && TS_CALLS=20 python tools/pmc_traffic.py ts $db "rocprofv3 --kernel-trace --pmc FETCH_SIZE -- python tools/pmc_traffic.py ts-run" profiles/r6_ts_encoder_pmc_fetch_size.txt )
rm -rf /tmp/fs5; C5="python bench.py --batch 16 --weights fp8 --series 8 --length 1024 --steps 4 --warmup 2 --no-cpu-baseline"
timeout 500 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d /tmp/fs5 -o p -- python $R/bench.py --batch 16 --weights fp8 --series 8 --length 1024 --steps 4 --warmup 2 --no-cpu-baseline > /tmp/fs5.log 2>&1
db=$(find /tmp/fs5 -name "*.db" | head -1)
(echo "## rocprofv3 --kernel-trace --pmc FETCH_SIZE -- $C5   (MI355X, round 6, final code)"; python $R/tools/prof_db.py $db | grep -v fill_hash | head -60) > $O/r6_cfg5_pmc_fetch_size.txt
( cd $R && python tools/pmc_traffic.py batched $db "rocprofv3 --kernel-trace --pmc FETCH_SIZE -- $C5" profiles/r6_cfg5_pmc_fetch_size.txt )
cd $R
timeout 900 python tools/parity_full_depth.py --model chatts-14b --out $O/r6_parity_14b_8x256_bf16_b1_full.json > $O/parity_14b.log 2>&1; echo "parity 14b rc=$?"; tail -2 $O/parity_14b.log | cut -c1-300
timeout 900 python tools/parity_full_depth.py --model chatts-8b --series 1 --length 256 --out $O/r6_parity_8b_1x256_bf16_b1_full.json > $O/parity_8b.log 2>&1; echo "parity 8b rc=$?"; tail -1 $O/parity_8b.log | cut -c1-300
timeout 900 python tools/parity_full_depth.py --series 30 --lengths mixed --out $O/r6_parity_14b_30xmixed_bf16_b1_full.json > $O/parity_cfg4.log 2>&1; echo "parity cfg4 rc=$?"; tail -1 $O/parity_cfg4.log | cut -c1-300
timeout 1200 python tools/parity_full_depth.py --series 8 --length 1024 --batch 16 --weights fp8 --oracle-slots 0,7,15 --out $O/r6_parity_14b_8x1024_fp8_b16_full.json > $O/parity_cfg5.log 2>&1; echo "parity cfg5 rc=$?"; tail -1 $O/parity_cfg5.log | cut -c1-300
timeout 900 python tools/parity_full_depth.py --model chatts-14b --precision f16q --out $O/r6_parity_14b_8x256_bf16_b1_f16q_full.json > $O/parity_f16q.log 2>&1; echo "parity f16q rc=$?"; tail -1 $O/parity_f16q.log | cut -c1-300
for wf in fp8 int8 int4; do
  timeout 900 python tools/parity_full_depth.py --model chatts-14b --weights $wf --out $O/r6_parity_14b_8x256_${wf}_b1_full.json > $O/parity_$wf.log 2>&1; echo "parity $wf rc=$?"; tail -1 $O/parity_$wf.log | cut -c1-300
done
# the bench lines read the parity files from profiles/: put this run's there for the lines below (the job's copies are what gets committed)
cp $O/r6_parity_*_full.json profiles/ 2>/dev/null
timeout 400 python bench.py --batch 16 --weights fp8 --series 8 --length 1024 --steps 24 --warmup 6 --no-cpu-baseline > $O/r6_bench_cfg5_fp8_8x1024_b16.json 2> $O/cfg5.err
timeout 400 python bench.py --series 30 --lengths mixed --steps 32 --warmup 8 --no-cpu-baseline > $O/r6_bench_cfg4_30xmixed.json 2> $O/cfg4.err
timeout 300 python bench.py --model chatts-8b --series 1 --length 256 --steps 32 --warmup 8 --no-cpu-baseline > $O/r6_bench_8b_cfg2.json 2> $O/8b.err
timeout 300 python bench.py --weights fp8 --steps 32 --warmup 8 --no-cpu-baseline > $O/r6_bench_fp8_weights.json 2> $O/fp8.err
timeout 300 python bench.py --weights int8 --steps 32 --warmup 8 --no-cpu-baseline > $O/r6_bench_int8_weights.json 2> $O/int8.err
timeout 300 python bench.py --weights int4 --steps 32 --warmup 8 --no-cpu-baseline > $O/r6_bench_int4_weights.json 2> $O/int4.err
timeout 300 python bench.py --precision f16q --steps 32 --warmup 8 --no-cpu-baseline > $O/r6_bench_f16q.json 2> $O/f16q.err
CHATTS_TILED_WEIGHTS=0 timeout 300 python bench.py --steps 32 --warmup 8 --no-cpu-baseline > $O/r6_bench_n1_rowmajor_weights.json 2> $O/rowmajor.err
timeout 600 python tools/tp_shard_step.py --worlds 1,2,4,8 --out $O/r6_tp_shard_step.json > /dev/null 2> $O/tp_shard_step.err
CHATTS_TP_BULK_FENCE=1 timeout 300 python tools/tp_shard_step.py --worlds 8 --out $O/r6_tp_shard_step_w8_threadfence.json > /dev/null 2> $O/tp_shard_step_fence.err
rm -rf /tmp/kt8
timeout 300 rocprofv3 --kernel-trace -d /tmp/kt8 -o p -- python tools/tp_shard_step.py --worlds 8 --steps 8 --out $O/tp8_traced.json > $O/tp8_trace.log 2>&1
(echo "## rocprofv3 --kernel-trace -- python tools/tp_shard_step.py --worlds 8 --steps 8   (one rank of TP = 8, loop-back exchange, MI355X, round 6, final code)"; python tools/prof_db.py $(find /tmp/kt8 -name "*.db" | head -1)) > $O/r6_tp8_shard_kernel_trace.txt
timeout 400 python tools/tp_shard_step.py --worlds 8 --batch 16 --weights fp8 --prefill-runs 1 --out $O/r6_tp8_shard_step_cfg5_batched.json > /dev/null 2> $O/tp8_cfg5.err
for f in r6_bench_n1 r6_bench_f16q r6_bench_n1_rowmajor_weights r6_bench_cfg5_fp8_8x1024_b16 r6_bench_cfg4_30xmixed r6_bench_8b_cfg2 r6_bench_fp8_weights r6_bench_int8_weights r6_bench_int4_weights; do
  python - $O/$f.json <<'PY'
import json, sys
try:
    r = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print(sys.argv[1].split("/")[-1], "value %.1f" % r["value"], "ms/step %.3f" % r["ms_per_step"], "ttft", r.get("ttft_ms_p50"), "parity_checked", r.get("parity_checked"),
          "traffic", (r.get("roofline") or {}).get("traffic"))
except Exception as e:
    print(sys.argv[1], "ERR", e)
PY
done
timeout 1800 python -m pytest tests/ -q -m gpu --durations=15 > $O/pytest_gpu_full.txt 2>&1; tail -22 $O/pytest_gpu_full.txt
