#!/bin/bash
R="${GRAFT_REPO_ROOT:-.}"; cd $R; mkdir -p gpurun_out
timeout 300 python -m pytest tests/test_gpu_kernels.py -q -x -m gpu -k "post_norm" 2>&1 | tail -3
timeout 600 python -m pytest tests/test_gpu_e2e.py tests/test_gpu_parity_real_size.py -q -x -m gpu -k "batch or config5 or continuous or engine or fp8" 2>&1 | tail -3
for q in 0 1; do
CHATTS_POST_NORM_SMALL_M=$q timeout 300 python bench.py --batch 16 --weights fp8 --series 8 --length 1024 --steps 24 --warmup 6 --no-cpu-baseline > gpurun_out/r3_cfg5_pn$q.json 2> gpurun_out/r3_cfg5_pn$q.err
python - <<PY
import json
d=json.loads(open("gpurun_out/r3_cfg5_pn$q.json").read().strip().splitlines()[-1])
print("post-norm small-M $q tok/s", d["value"], "ms/step", d["ms_per_step"], "parity", d.get("parity_checked"))
PY
done
timeout 300 python bench.py --batch 16 --series 8 --length 256 --steps 24 --warmup 6 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('bf16 b16 8x256', d['value'], d['ms_per_step'])"
