#!/bin/bash
# persistent decode step, first contact: tiny models first (a hang costs least there), then 14B widths.  Everything under `timeout`.
R="${GRAFT_REPO_ROOT:-.}"; cd "$R"; mkdir -p gpurun_out
timeout 300 python -m pytest tests/test_gpu_decode_mega.py -x -q --timeout 120 -k "bit_identical or sampling" 2>&1 | tail -15 > gpurun_out/r3_mega1_pytest.log
tail -15 gpurun_out/r3_mega1_pytest.log
timeout 200 python tools/mega_check.py --layers 4 --steps 32 > gpurun_out/r3_mega1_l4.json 2> gpurun_out/r3_mega1_l4.err; echo "l4 rc=$?"; tail -c 800 gpurun_out/r3_mega1_l4.json; tail -3 gpurun_out/r3_mega1_l4.err
timeout 300 python tools/mega_check.py --steps 32 > gpurun_out/r3_mega1_full.json 2> gpurun_out/r3_mega1_full.err; echo "full rc=$?"; tail -c 800 gpurun_out/r3_mega1_full.json; tail -3 gpurun_out/r3_mega1_full.err
