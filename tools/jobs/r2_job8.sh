#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out
timeout 200 python -m pytest tests/test_gpu_tp_p2p.py -x -q --timeout 60 > gpurun_out/r2_job8_tp.log 2>&1
tail -8 gpurun_out/r2_job8_tp.log
timeout 120 python tools/tp_exchange_bench.py > gpurun_out/r2_tp_exchange_bench.txt 2>&1
cat gpurun_out/r2_tp_exchange_bench.txt
