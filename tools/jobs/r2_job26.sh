#!/bin/bash
mkdir -p gpurun_out
timeout 900 python -m pytest tests -m gpu -x -q --timeout 300 2>&1 | tail -4
for kb in 0 64; do
timeout 300 python bench.py --steps 16 --warmup 4 --no-cpu-baseline --ttft-runs 5 --kv-block $kb > gpurun_out/r2_bench_kvblock$kb.json 2> gpurun_out/r2_bench_kvblock$kb.err
done
for kb in 0 64; do
timeout 400 python bench.py --batch 16 --weights fp8 --length 1024 --max-ctx 2048 --steps 16 --warmup 4 --no-cpu-baseline --kv-block $kb > gpurun_out/r2_bench_cfg5_kvblock$kb.json 2> gpurun_out/r2_bench_cfg5_kvblock$kb.err
done
python - <<'PY'
import json
for f in ('r2_bench_kvblock0','r2_bench_kvblock64','r2_bench_cfg5_kvblock0','r2_bench_cfg5_kvblock64'):
    try:
        d=json.loads(open(f'gpurun_out/{f}.json').read().strip().splitlines()[-1])
        print(f, {k:d.get(k) for k in ('value','ttft_ms_p50','ms_per_step')}, d['config'].get('kv_block'))
    except Exception as e:
        print(f, 'failed', e, open(f'gpurun_out/{f}.err').read()[-600:])
PY
