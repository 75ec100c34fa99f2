#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_tp_p2p.py -x -q > gpurun_out/r2_tp_tests.log 2>&1
tail -15 gpurun_out/r2_tp_tests.log
timeout 600 python -m pytest tests/test_gpu_e2e.py -x -q > gpurun_out/r2_e2e_tests.log 2>&1
tail -5 gpurun_out/r2_e2e_tests.log
bash tools/jobs/tp2_single_device.sh
timeout 300 python bench.py --steps 32 --warmup 8 --no-cpu-baseline > gpurun_out/r2_bench_a.json 2> gpurun_out/r2_bench_a.err
tail -c 2500 gpurun_out/r2_bench_a.json
