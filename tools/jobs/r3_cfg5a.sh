#!/bin/bash
R="${GRAFT_REPO_ROOT:-.}"; cd "$R"; mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_decode_mega.py -x -q --timeout 300 -k "attention or decode or persistent" 2>&1 | tail -6
for NS in 8 16 32 64; do
  CHATTS_BATCH_NSPLITS=$NS timeout 300 python bench.py --batch 16 --weights fp8 --series 8 --length 1024 --steps 24 --warmup 6 --no-cpu-baseline > gpurun_out/r3_cfg5_ns$NS.json 2> gpurun_out/r3_cfg5_ns$NS.err
  NS=$NS python - <<'PY'
import json, os
d = json.loads(open("gpurun_out/r3_cfg5_ns%s.json" % os.environ["NS"]).read().strip().splitlines()[-1])
print("nsplits", os.environ["NS"], "tok/s", round(d["value"], 1), "ms/step", round(d["ms_per_step"], 3), "parity", d["parity_checked"])
PY
done
timeout 300 python bench.py --series 30 --lengths mixed --steps 24 --warmup 6 --no-cpu-baseline --ttft-runs 2 > gpurun_out/r3_cfg4_pf.json 2> gpurun_out/r3_cfg4_pf.err
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r3_cfg4_pf.json").read().strip().splitlines()[-1])
print("cfg4 tok/s", round(d["value"], 1), "ttft", round(d["ttft_ms_p50"], 1), "parity", d["parity_checked"])
PY
timeout 300 python bench.py --steps 24 --warmup 6 --no-cpu-baseline --ttft-runs 2 > gpurun_out/r3_cfg3_pf.json 2> gpurun_out/r3_cfg3_pf.err
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r3_cfg3_pf.json").read().strip().splitlines()[-1])
print("cfg3 tok/s", round(d["value"], 1), "ttft", round(d["ttft_ms_p50"], 1), "parity", d["parity_checked"])
PY
