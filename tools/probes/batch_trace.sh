cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
timeout 400 rocprofv3 --kernel-trace -d /tmp/bt -o p -- python $R/tools/batch_bench.py 16 > /tmp/bt.log 2>&1
db=$(find /tmp/bt -name "*.db" | head -1)
python $R/tools/prof_db.py $db > $R/gpurun_out/batch16_trace.txt
tail -2 /tmp/bt.log
