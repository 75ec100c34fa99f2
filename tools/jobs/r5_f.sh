#!/bin/bash
# round 5, call F: the whole GPU suite after the clean-up (options table, old prefill kernel and the twice-lost variants removed),
# with durations, then the bench
set -x
mkdir -p gpurun_out/r5_f
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests/ -q -m gpu --durations=60 -x > gpurun_out/r5_f/pytest_gpu.txt 2>&1
tail -80 gpurun_out/r5_f/pytest_gpu.txt
timeout 600 python bench.py --steps 16 --warmup 2 --no-cpu-baseline --ttft-runs 5 > gpurun_out/r5_f/bench.txt 2>&1
tail -1 gpurun_out/r5_f/bench.txt | cut -c1-600
