cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
timeout 300 rocprofv3 --kernel-trace -d /tmp/kt -o p -- python $R/bench.py --steps 8 --warmup 2 --no-cpu-baseline --ttft-runs 3 > /tmp/kt.log 2>&1
db=$(find /tmp/kt -name "*.db" | head -1)
python $R/tools/prof_db.py $db > $R/gpurun_out/kt_prefill.txt
tail -2 /tmp/kt.log | cut -c1-300
