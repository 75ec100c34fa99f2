#!/bin/bash
# round-2 final evidence: the bench line as the driver runs it (with the CPU baseline), then the rocprofv3 kernel trace of the same
# command shape; summaries are copied into profiles/ by hand afterwards
R="${GRAFT_REPO_ROOT:-.}"
cd /tmp && export TMPDIR=/tmp
mkdir -p $R/gpurun_out
( cd $R && timeout 400 python bench.py --steps 32 --warmup 8 > gpurun_out/r2_bench_n1.json 2> gpurun_out/r2_bench_n1.err )
rm -rf /tmp/kt
timeout 300 rocprofv3 --kernel-trace -d /tmp/kt -o p -- python $R/bench.py --steps 8 --warmup 2 --no-cpu-baseline --ttft-runs 3 > /tmp/kt.log 2>&1
db=$(find /tmp/kt -name "*.db" | head -1)
(echo "## rocprofv3 --kernel-trace -- python bench.py --steps 8 --warmup 2 --no-cpu-baseline --ttft-runs 3   (MI355X, round 2, final code)"; python $R/tools/prof_db.py $db) > $R/gpurun_out/r2_bench_kernel_trace.txt
python - <<PY
import json
d=json.loads(open("$R/gpurun_out/r2_bench_n1.json").read().strip().splitlines()[-1])
print({k:d[k] for k in ('value','ttft_ms_p50','ts_encode_ms_p50','parity_checked')}, d['roofline']['frac'], d['roofline']['avg_us'], d['cpu_baseline']['value'])
PY
head -25 $R/gpurun_out/r2_bench_kernel_trace.txt | cut -c1-170
