#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out
timeout 300 python -m pytest tests/test_gpu_kernels.py -x -q --timeout 60 -k "int4 or normalise" > gpurun_out/r2_job10_k.log 2>&1
tail -15 gpurun_out/r2_job10_k.log
timeout 300 python -m pytest tests/test_gptq.py tests/test_gpu_e2e.py -x -q --timeout 150 -k "gptq or int4 or fp8" > gpurun_out/r2_job10_e.log 2>&1
tail -15 gpurun_out/r2_job10_e.log
timeout 300 python bench.py --weights int4 --steps 32 --warmup 8 --no-cpu-baseline --ttft-runs 2 > gpurun_out/r2_bench_int4.json 2> gpurun_out/r2_bench_int4.err
python - <<PY
import json
d = json.loads(open("gpurun_out/r2_bench_int4.json").read().strip().splitlines()[-1])
print("int4 tok/s", round(d["value"], 2), "ms", round(d["ms_per_step"], 4), d["roofline"]["kernel"], round(d["roofline"]["achieved"]), d["roofline"]["avg_us"])
PY
tail -3 gpurun_out/r2_bench_int4.err
