#!/bin/bash
# round-3 final evidence: the bench line as the driver runs it (with the CPU baseline), the rocprofv3 kernel trace of the same command
# shape, the config-4 / config-5 / 8B / int4 lines; summaries are copied into profiles/ by hand afterwards
R="${GRAFT_REPO_ROOT:-.}"
cd /tmp && export TMPDIR=/tmp
mkdir -p $R/gpurun_out
( cd $R && timeout 500 python bench.py --steps 32 --warmup 8 > gpurun_out/r3_bench_n1.json 2> gpurun_out/r3_bench_n1.err )
rm -rf /tmp/kt
timeout 300 rocprofv3 --kernel-trace -d /tmp/kt -o p -- python $R/bench.py --steps 8 --warmup 2 --no-cpu-baseline --ttft-runs 3 > /tmp/kt.log 2>&1
db=$(find /tmp/kt -name "*.db" | head -1)
(echo "## rocprofv3 --kernel-trace -- python bench.py --steps 8 --warmup 2 --no-cpu-baseline --ttft-runs 3   (MI355X, round 3, final code)"; python $R/tools/prof_db.py $db) > $R/gpurun_out/r3_bench_kernel_trace.txt
rm -rf /tmp/kt5
timeout 400 rocprofv3 --kernel-trace -d /tmp/kt5 -o p -- python $R/bench.py --batch 16 --weights fp8 --series 8 --length 1024 --steps 8 --warmup 2 --no-cpu-baseline > /tmp/kt5.log 2>&1
db=$(find /tmp/kt5 -name "*.db" | head -1)
(echo "## rocprofv3 --kernel-trace -- python bench.py --batch 16 --weights fp8 --series 8 --length 1024 --steps 8 --warmup 2 --no-cpu-baseline  (MI355X, round 3, final code)"; python $R/tools/prof_db.py $db) > $R/gpurun_out/r3_cfg5_kernel_trace.txt
cd $R
timeout 400 python bench.py --batch 16 --weights fp8 --series 8 --length 1024 --steps 24 --warmup 6 --no-cpu-baseline > gpurun_out/r3_bench_cfg5_fp8_8x1024_b16.json 2> gpurun_out/r3_bench_cfg5.err
timeout 400 python bench.py --series 30 --lengths mixed --steps 32 --warmup 8 --no-cpu-baseline > gpurun_out/r3_bench_cfg4_30xmixed.json 2> gpurun_out/r3_bench_cfg4.err
timeout 300 python bench.py --model chatts-8b --series 1 --length 256 --steps 32 --warmup 8 --no-cpu-baseline > gpurun_out/r3_bench_8b_cfg2.json 2> gpurun_out/r3_bench_8b.err
timeout 300 python bench.py --weights int4 --steps 32 --warmup 8 --no-cpu-baseline > gpurun_out/r3_bench_int4_weights.json 2> gpurun_out/r3_bench_int4.err
timeout 300 python bench.py --weights fp8 --steps 32 --warmup 8 --no-cpu-baseline > gpurun_out/r3_bench_fp8_weights.json 2> gpurun_out/r3_bench_fp8.err
python - <<PY
import json
for f in ("r3_bench_n1","r3_bench_cfg5_fp8_8x1024_b16","r3_bench_cfg4_30xmixed","r3_bench_8b_cfg2","r3_bench_int4_weights","r3_bench_fp8_weights"):
    try:
        d=json.loads(open("gpurun_out/%s.json"%f).read().strip().splitlines()[-1])
        print(f, round(d["value"],1), round(d["ms_per_step"],3), d.get("ttft_ms_p50"), d.get("parity_checked"), round(d["roofline"]["frac"],3), d["roofline"].get("traffic"), (d.get("cpu_baseline") or {}).get("value"), (d.get("cpu_baseline") or {}).get("ts_encode_ms"))
    except Exception as e:
        print(f, "FAILED", e)
PY
head -24 gpurun_out/r3_bench_kernel_trace.txt | cut -c1-170
