#!/bin/bash
# full GPU suite on the code with the planes attention, then the A/B bench lines
cd /root/repo; mkdir -p gpurun_out
python -m pytest tests -x -q -m gpu 2>&1 | tail -4 | tee gpurun_out/r4_v_suite.txt
python bench.py --steps 16 --warmup 4 --no-cpu-baseline --ttft-runs 7 2>/dev/null | tail -1 > gpurun_out/r4_v_bench.json
CHATTS_ATTN_PLANES=0 python bench.py --steps 16 --warmup 4 --no-cpu-baseline --ttft-runs 7 2>/dev/null | tail -1 > gpurun_out/r4_v_bench_noplanes.json
python -c "
import json
for f in ('r4_v_bench','r4_v_bench_noplanes'):
    r=json.load(open('gpurun_out/%s.json'%f)); print(f, {k: r.get(k) for k in ('value','ms_per_step','ttft_ms_p50','parity_checked')})"
