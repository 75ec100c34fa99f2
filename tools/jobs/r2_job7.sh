#!/bin/bash
# round-2 profiles: kernel trace + FETCH_SIZE pass of the bench command, exchange-kernel latency
R="${GRAFT_REPO_ROOT:-.}"
cd /tmp && export TMPDIR=/tmp
mkdir -p $R/gpurun_out
timeout 300 rocprofv3 --kernel-trace -d /tmp/kt -o p -- python $R/bench.py --steps 8 --warmup 2 --no-cpu-baseline --ttft-runs 3 > /tmp/kt.log 2>&1
db=$(find /tmp/kt -name "*.db" | head -1)
(echo "## rocprofv3 --kernel-trace -- python bench.py --steps 8 --warmup 2 --no-cpu-baseline --ttft-runs 3   (MI355X, round 2)"; python $R/tools/prof_db.py $db) > $R/gpurun_out/r2_bench_kernel_trace.txt
tail -2 /tmp/kt.log | cut -c1-200
timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d /tmp/fs -o p -- python $R/bench.py --steps 4 --warmup 2 --no-cpu-baseline --ttft-runs 1 > /tmp/fs.log 2>&1
db=$(find /tmp/fs -name "*.db" | head -1)
(echo "## rocprofv3 --kernel-trace --pmc FETCH_SIZE -- python bench.py --steps 4 --warmup 2 --no-cpu-baseline --ttft-runs 1   (MI355X, round 2)"; python $R/tools/prof_db.py $db) > $R/gpurun_out/r2_bench_pmc_fetch_size.txt
tail -2 /tmp/fs.log | cut -c1-200
cd $R
timeout 120 python tools/tp_exchange_bench.py > gpurun_out/r2_tp_exchange_bench.txt 2>&1
cat gpurun_out/r2_tp_exchange_bench.txt
head -60 gpurun_out/r2_bench_kernel_trace.txt
