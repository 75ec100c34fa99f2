#!/bin/bash
mkdir -p gpurun_out
timeout 900 python -m pytest tests -m gpu -x -q --timeout 300 2>&1 | tail -5
timeout 120 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
timeout 300 python tools/attn_prefill_time.py 2>&1 | tail -8
timeout 300 python bench.py --steps 16 --warmup 4 --no-cpu-baseline --ttft-runs 7 > gpurun_out/r2_bench_f.json 2> gpurun_out/r2_bench_f.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r2_bench_f.json').read().strip().splitlines()[-1])
print({k:d[k] for k in ('value','ttft_ms_p50','ts_encode_ms_p50','parity_checked')}, d['prefill_roofline']['avg_us'])
PY
