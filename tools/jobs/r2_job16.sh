#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out
timeout 400 python -m pytest tests/test_gpu_e2e.py tests/test_gpu_parity_real_size.py tests/test_gpu_server.py -x -q --timeout 200 > gpurun_out/r2_job16.log 2>&1
tail -8 gpurun_out/r2_job16.log
timeout 200 python bench.py --steps 8 --warmup 2 --no-cpu-baseline --ttft-runs 5 > gpurun_out/r2_bench_last.json 2> gpurun_out/r2_bench_last.err
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r2_bench_last.json").read().strip().splitlines()[-1])
print("ttft", round(d["ttft_ms_p50"], 3), "tok/s", round(d["value"], 1), "parity", d["parity_checked"], "ts", round(d["ts_encoder_roofline"]["avg_us"], 1))
PY
