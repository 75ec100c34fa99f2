# Round evidence: the bench line as the driver runs it + a kernel-trace summary of the same command.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
python $R/bench.py --gpus 1 --steps 64 --warmup 8 > /tmp/bench.log 2>&1
tail -1 /tmp/bench.log > $R/gpurun_out/bench_n1.json
timeout 400 rocprofv3 --kernel-trace --stats -d /tmp/kt -o p -- python $R/bench.py --gpus 1 --steps 64 --warmup 8 --no-cpu-baseline > /tmp/kt.log 2>&1
db=$(find /tmp/kt -name "*.db" | head -1)
python $R/tools/prof_db.py $db > $R/gpurun_out/bench_kernel_trace.txt
ls /tmp/kt | head; find /tmp/kt -name "*stats*" | head
for f in $(find /tmp/kt -name "*kernel_stats*.csv" | head -1); do head -40 $f > $R/gpurun_out/bench_kernel_stats.csv; done
tail -1 /tmp/kt.log | cut -c1-400
