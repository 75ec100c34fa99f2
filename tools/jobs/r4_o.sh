#!/bin/bash
R="${GRAFT_REPO_ROOT:-.}"; O=$R/gpurun_out; mkdir -p $O
cd /tmp && export TMPDIR=/tmp; rm -rf /tmp/kt8
timeout 600 rocprofv3 --kernel-trace -d /tmp/kt8 -o p -- python $R/tools/tp_shard_step.py --worlds 8 --batch 16 --weights fp8 --steps 8 --warmup 2 --prefill-runs 1 --out $O/r4o.json > /tmp/kt8.log 2>&1
(echo "## rocprofv3 --kernel-trace -- python tools/tp_shard_step.py --worlds 8 --batch 16 --weights fp8 --steps 8 --warmup 2 --prefill-runs 1  (ONE rank of TP=8, 16-wide decode step at ctx 1207, loop-back exchange, MI355X, round 4)"; python $R/tools/prof_db.py $(find /tmp/kt8 -name "*.db" | head -1) | grep -v fill_hash) > $O/r4_tp8_cfg5_shard_kernel_trace.txt
head -30 $O/r4_tp8_cfg5_shard_kernel_trace.txt | cut -c1-200
