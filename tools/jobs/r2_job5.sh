#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_server.py tests/test_gpu_e2e.py tests/test_gpu_tp_p2p.py -x -q > gpurun_out/r2_job5_tests.log 2>&1
tail -15 gpurun_out/r2_job5_tests.log
timeout 300 python -m pytest tests/test_gpu_kernels.py -x -q -k "ts_ or gemm" > gpurun_out/r2_job5_kernels.log 2>&1
tail -5 gpurun_out/r2_job5_kernels.log
timeout 400 python bench.py --steps 32 --warmup 8 > gpurun_out/r2_bench_c.json 2> gpurun_out/r2_bench_c.err
tail -c 1200 gpurun_out/r2_bench_c.json; tail -3 gpurun_out/r2_bench_c.err
timeout 400 python bench.py --weights fp8 --length 1024 --batch 16 --max-ctx 2048 --steps 32 --warmup 8 > gpurun_out/r2_bench_cfg5.json 2> gpurun_out/r2_bench_cfg5.err
head -c 1500 gpurun_out/r2_bench_cfg5.json; tail -3 gpurun_out/r2_bench_cfg5.err
timeout 300 python bench.py --model chatts-8b --series 1 --steps 32 --warmup 8 --no-cpu-baseline > gpurun_out/r2_bench_8b_cfg2.json 2> gpurun_out/r2_bench_8b_cfg2.err
head -c 900 gpurun_out/r2_bench_8b_cfg2.json
