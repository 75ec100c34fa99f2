#!/bin/bash
# round 4, call L: the whole GPU suite on the current code (+ the per-rank step with the gate_up K-split)
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out; O=gpurun_out
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 600 python tools/tp_shard_step.py --worlds 4,8 --prefill-runs 1 --out $O/r4l_shard.json > /dev/null 2> $O/r4l_shard.err; grep tp_shard_step $O/r4l_shard.err | cut -c1-250
timeout 2400 python -m pytest tests/ -x -q -m gpu > $O/r4l_gpu_suite.log 2>&1; echo "gpu suite rc=$?"; tail -8 $O/r4l_gpu_suite.log | cut -c1-300
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
