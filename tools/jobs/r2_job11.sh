#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out
timeout 300 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_e2e.py -x -q --timeout 100 -k "int4" > gpurun_out/r2_job11.log 2>&1
tail -5 gpurun_out/r2_job11.log
for OCC in default 1 3; do
if [ $OCC = default ]; then unset CHATTS_GEMV_OCC; else export CHATTS_GEMV_OCC=$OCC; fi
timeout 300 python bench.py --weights int4 --steps 32 --warmup 8 --no-cpu-baseline --ttft-runs 1 > gpurun_out/r2_bench_int4.json 2> gpurun_out/r2_bench_int4.err
python - <<PY
import json
d = json.loads(open("gpurun_out/r2_bench_int4.json").read().strip().splitlines()[-1])
print("OCC=$OCC int4 tok/s", round(d["value"], 2), "ms", round(d["ms_per_step"], 4), round(d["roofline"]["achieved"]), round(d["roofline"]["avg_us"],2))
PY
done
