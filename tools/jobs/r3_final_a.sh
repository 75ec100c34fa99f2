#!/bin/bash
R="${GRAFT_REPO_ROOT:-.}"; cd $R; mkdir -p gpurun_out
timeout 120 tools/probes/row_stream_probe 2>&1 | tee gpurun_out/r3_row_stream_probe.txt
timeout 300 python -m pytest tests/test_gpu_kernels.py -q -x -m gpu -k "argmax" 2>&1 | tail -3
timeout 400 python bench.py --steps 32 --warmup 8 --no-cpu-baseline > gpurun_out/r3_bench_quick.json 2> gpurun_out/r3_bench_quick.err
python - <<PY
import json
d=json.loads(open("gpurun_out/r3_bench_quick.json").read().strip().splitlines()[-1])
print("n1", d["value"], d["ms_per_step"], d.get("ttft_ms_p50"), d["parity_checked"], d["roofline"]["frac"])
PY
