#!/bin/bash
# round 4, call D: two-shot bulk all-reduce + prefill under TP as one C call; per-rank step and prefill (loop-back) again
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out; O=gpurun_out; R=$PWD
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 1200 python -m pytest tests/test_gpu_tp_p2p.py tests/test_gpu_tp_multiprocess.py -m gpu -q -s -x > $O/r4d_tests.log 2>&1; echo "tests rc=$?"; tail -30 $O/r4d_tests.log | cut -c1-900
timeout 600 python tools/tp_shard_step.py --worlds 2,4,8 --out $O/r4d_shard.json > /dev/null 2> $O/r4d_shard.err; grep tp_shard_step $O/r4d_shard.err
cd /tmp && export TMPDIR=/tmp; rm -rf /tmp/kt8
timeout 600 rocprofv3 --kernel-trace -d /tmp/kt8 -o p -- python $R/tools/tp_shard_step.py --worlds 8 --steps 8 --warmup 2 --prefill-runs 2 --out $R/$O/r4_tp8_traced.json > /tmp/kt8.log 2>&1; echo "rocprof rc=$?"
db=$(find /tmp/kt8 -name "*.db" | head -1)
(echo "## rocprofv3 --kernel-trace -- python tools/tp_shard_step.py --worlds 8 --steps 8 --warmup 2 --prefill-runs 2   (ONE rank of TP=8, loop-back exchange, MI355X, round 4)"; python $R/tools/prof_db.py $db) > $R/$O/r4d_tp8_shard_kernel_trace.txt
cd $R; head -24 $O/r4d_tp8_shard_kernel_trace.txt | cut -c1-200
