#!/bin/bash
# last-row attention split + inference drivers + TTFT without the mid-region sync
mkdir -p gpurun_out
timeout 500 python -m pytest tests/test_gpu_e2e.py tests/test_gpu_parity_real_size.py -x -q --timeout 200 2>&1 | tail -5
timeout 300 python bench.py --steps 16 --warmup 4 --no-cpu-baseline --ttft-runs 7 > gpurun_out/r2_bench_e.json 2> gpurun_out/r2_bench_e.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r2_bench_e.json').read().strip().splitlines()[-1])
print({k:d[k] for k in ('value','ttft_ms_p50','ts_encode_ms_p50','parity_checked')})
PY
