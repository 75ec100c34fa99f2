#!/bin/bash
# round-6 final evidence, second pass (after the fused TS layer 0, the 8-wave fp8 stream default): bench lines, traces, FETCH_SIZE passes
# with code digests; the full-depth parity files of tools/jobs/r6_final.sh stay valid (no projection / attention arithmetic changed since).
# round-6 final evidence on the final code: the bench line as the driver runs it, kernel traces, FETCH_SIZE passes with code digests
# (-> gpurun_out/r6_final2/pmc_traffic.json, copied to profiles/ afterwards), full-depth parity of every workload that has a bench line,
# the other workloads' bench lines, the per-rank TP steps.  Summaries are copied into profiles/ by hand afterwards.
R="${GRAFT_REPO_ROOT:-.}"; O=$R/gpurun_out/r6_final2; mkdir -p $O
export PMC_TRAFFIC_OUT=$O/pmc_traffic.json HSA_ENABLE_IPC_MODE_LEGACY=0
cp $R/profiles/pmc_traffic.json $O/pmc_traffic.json
cd /tmp && export TMPDIR=/tmp
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
cp $O/pmc_traffic.json profiles/pmc_traffic.json      # (on the box: the bench lines below quote the counter passes just collected - same sources)
( cd $R && timeout 600 python bench.py --steps 32 --warmup 8 > $O/r6_bench_n1.json 2> $O/r6_bench_n1.err ); echo "bench rc=$?"
# the bench lines read the parity files from profiles/: put this run's there for the lines below (the job's copies are what gets committed)
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
