#!/bin/bash
# final per-rank TP numbers on the final exchange code + the whole GPU suite
R="${GRAFT_REPO_ROOT:-.}"; O=$R/gpurun_out; mkdir -p $O; cd $R
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 600 python tools/tp_shard_step.py --worlds 1,2,4,8 --out $O/r4_tp_shard_step.json > /dev/null 2> $O/r4_tp_shard_step.err; grep tp_shard_step $O/r4_tp_shard_step.err | cut -c1-230
timeout 400 python tools/tp_shard_step.py --worlds 8 --batch 16 --weights fp8 --prefill-runs 1 --out $O/r4_tp8_shard_step_cfg5.json > /dev/null 2> $O/r4_tp8_shard_step_cfg5.err; grep tp_shard_step $O/r4_tp8_shard_step_cfg5.err | cut -c1-260
cd /tmp && export TMPDIR=/tmp; rm -rf /tmp/kt8
timeout 600 rocprofv3 --kernel-trace -d /tmp/kt8 -o p -- python $R/tools/tp_shard_step.py --worlds 8 --steps 8 --warmup 2 --prefill-runs 2 --out $O/r4_tp8_traced.json > /tmp/kt8.log 2>&1
(echo "## rocprofv3 --kernel-trace -- python tools/tp_shard_step.py --worlds 8 --steps 8 --warmup 2 --prefill-runs 2   (ONE rank of TP=8, loop-back exchange, MI355X, round 4, final code)"; python $R/tools/prof_db.py $(find /tmp/kt8 -name "*.db" | head -1)) > $O/r4_tp8_shard_kernel_trace.txt
rm -rf /tmp/kt8b
timeout 600 rocprofv3 --kernel-trace -d /tmp/kt8b -o p -- python $R/tools/tp_shard_step.py --worlds 8 --batch 16 --weights fp8 --steps 8 --warmup 2 --prefill-runs 1 --out $O/r4o.json > /tmp/kt8b.log 2>&1
(echo "## rocprofv3 --kernel-trace -- python tools/tp_shard_step.py --worlds 8 --batch 16 --weights fp8 --steps 8 --warmup 2 --prefill-runs 1  (ONE rank of TP=8, 16-wide decode step at ctx 1207, loop-back exchange, MI355X, round 4, final code)"; python $R/tools/prof_db.py $(find /tmp/kt8b -name "*.db" | head -1) | grep -v fill_hash) > $O/r4_tp8_cfg5_shard_kernel_trace.txt
cd $R; grep "tp_allreduce" $O/r4_tp8_shard_kernel_trace.txt $O/r4_tp8_cfg5_shard_kernel_trace.txt | cut -c1-200
timeout 2400 python -m pytest tests/ -q -m gpu > $O/r4q_gpu_suite.log 2>&1; echo "gpu suite rc=$?"; tail -6 $O/r4q_gpu_suite.log | cut -c1-300
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
