#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out
timeout 500 python -m pytest tests/test_gpu_e2e.py tests/test_gpu_parity_real_size.py -x -q --timeout 200 > gpurun_out/r2_job20.log 2>&1
tail -6 gpurun_out/r2_job20.log
timeout 200 python bench.py --steps 8 --warmup 2 --no-cpu-baseline --ttft-runs 5 > gpurun_out/r2_bench_x3.json 2> gpurun_out/r2_bench_x3.err
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r2_bench_x3.json").read().strip().splitlines()[-1])
print("ttft", round(d["ttft_ms_p50"], 3), "tok/s", round(d["value"], 1), "parity", d["parity_checked"])
PY
timeout 300 python tools/parity_full_depth.py --model chatts-14b --out gpurun_out/r2_parity_14b_full.json > gpurun_out/r2_parity_14b_full.log 2>&1
tail -c 700 gpurun_out/r2_parity_14b_full.log
