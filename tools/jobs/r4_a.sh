#!/bin/bash
# round 4, call A: parity of the TP shard shapes (W = 8 / 4 at 14B widths), the exchange inside the GEMV launch (W = 2, two in-process
# ranks), and the first MEASURED per-rank step times at the shard shapes (loop-back exchange), fused vs stand-alone exchange.
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out; O=gpurun_out
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 900 python -m pytest tests/test_gpu_tp_shards.py tests/test_gpu_tp_p2p.py -m gpu -x -q -s > $O/r4a_tests.log 2>&1; echo "tests rc=$?"; tail -15 $O/r4a_tests.log
timeout 600 python tools/tp_shard_step.py --worlds 1,2,4,8 --out $O/r4_tp_shard_step_fused.json > /dev/null 2> $O/r4a_shard_fused.err; echo "shard fused rc=$?"; grep tp_shard_step $O/r4a_shard_fused.err
CHATTS_TP_FUSE=0 timeout 600 python tools/tp_shard_step.py --worlds 2,8 --out $O/r4_tp_shard_step_standalone.json > /dev/null 2> $O/r4a_shard_sa.err; echo "shard standalone rc=$?"; grep tp_shard_step $O/r4a_shard_sa.err
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; rm -rf /tmp/kt8
timeout 600 rocprofv3 --kernel-trace -d /tmp/kt8 -o p -- python $R/tools/tp_shard_step.py --worlds 8 --steps 8 --warmup 2 --prefill-runs 1 --out $R/$O/r4_tp8_traced.json > /tmp/kt8.log 2>&1; echo "rocprof rc=$?"
db=$(find /tmp/kt8 -name "*.db" | head -1)
(echo "## rocprofv3 --kernel-trace -- python tools/tp_shard_step.py --worlds 8 --steps 8 --warmup 2 --prefill-runs 1   (ONE rank of TP=8, loop-back exchange, MI355X, round 4)"; python $R/tools/prof_db.py $db) > $R/$O/r4_tp8_shard_kernel_trace.txt
cd $R; head -45 $O/r4_tp8_shard_kernel_trace.txt
timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/r4a_bench.json 2> $O/r4a_bench.err; echo "bench rc=$?"; head -c 600 $O/r4a_bench.json
