#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out
export TMPDIR=/tmp
(time timeout 1500 python -m pytest tests -x -q -m gpu --timeout 300) > gpurun_out/r2_gpu_suite.log 2>&1
tail -6 gpurun_out/r2_gpu_suite.log
timeout 300 python bench.py --steps 32 --warmup 8 --ttft-runs 5 > gpurun_out/r2_bench_d.json 2> gpurun_out/r2_bench_d.err
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r2_bench_d.json").read().strip().splitlines()[-1])
print("ttft", round(d["ttft_ms_p50"], 3), "tok/s", round(d["value"], 1), "parity", d["parity_checked"], "cpu", d["cpu_baseline"]["value"], d["cpu_baseline"]["wall_s"])
PY
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
