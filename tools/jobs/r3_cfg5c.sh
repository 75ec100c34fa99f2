#!/bin/bash
R="${GRAFT_REPO_ROOT:-.}"; cd $R; mkdir -p gpurun_out
timeout 300 python -m pytest tests/test_gpu_kernels.py -q -x -m gpu -k "gemm_stream or split_k" 2>&1 | tail -5
timeout 300 python -m pytest tests/test_gpu_batched.py tests/test_gpu_parity_real_size.py -q -x -m gpu 2>&1 | tail -3
for fx in 0 1; do
CHATTS_GEMM_FIXUP=$fx timeout 300 python bench.py --batch 16 --weights fp8 --series 8 --length 1024 --steps 24 --warmup 6 --no-cpu-baseline > gpurun_out/r3_cfg5_fix$fx.json 2> gpurun_out/r3_cfg5_fix$fx.err
python - <<PY
import json
d=json.loads(open("gpurun_out/r3_cfg5_fix$fx.json").read().strip().splitlines()[-1])
print("fixup $fx tok/s", d["value"], "ms/step", d["ms_per_step"], "parity", d.get("parity_checked"))
PY
done
