#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out
timeout 300 python -m pytest tests/test_gpu_kernels.py -x -q --timeout 60 -k "attention_decode" > gpurun_out/r2_job9_attn.log 2>&1
tail -12 gpurun_out/r2_job9_attn.log
timeout 400 python -m pytest tests/test_gpu_e2e.py tests/test_gpu_parity_real_size.py -x -q --timeout 200 > gpurun_out/r2_job9_e2e.log 2>&1
tail -8 gpurun_out/r2_job9_e2e.log
for P in -1 0; do
CHATTS_ATTN_PARTS=$P timeout 200 python bench.py --steps 32 --warmup 8 --no-cpu-baseline --ttft-runs 2 > gpurun_out/r2_bench_parts$P.json 2> gpurun_out/r2_bench_parts$P.err
python - <<PY
import json
d = json.loads(open("gpurun_out/r2_bench_parts$P.json").read().strip().splitlines()[-1])
print("PARTS=$P tok/s", round(d["value"], 2), "ms", round(d["ms_per_step"], 4), "parity", d["parity_checked"])
PY
done
