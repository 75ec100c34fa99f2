#!/bin/bash
R="${GRAFT_REPO_ROOT:-.}"; cd $R; mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_e2e.py -q -x -m gpu -k "int8 or fp8" 2>&1 | tail -4
timeout 300 python bench.py --weights int8 --steps 32 --warmup 8 --no-cpu-baseline > gpurun_out/r3_bench_int8_weights.json 2> gpurun_out/r3_bench_int8.err
timeout 300 python bench.py --batch 16 --weights int8 --series 8 --length 1024 --steps 16 --warmup 4 --no-cpu-baseline > gpurun_out/r3_bench_int8_b16.json 2> gpurun_out/r3_bench_int8_b16.err
python - <<PY
import json
for f in ("r3_bench_int8_weights","r3_bench_int8_b16"):
    try:
        d=json.loads(open("gpurun_out/%s.json"%f).read().strip().splitlines()[-1]); print(f, d["value"], d["ms_per_step"], d["roofline"]["frac"], d["dtype"][:40])
    except Exception as e:
        print(f, "FAILED", e); print(open("gpurun_out/%s.err"%f.replace("_weights","")).read()[-800:])
PY
