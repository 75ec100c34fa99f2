#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out
timeout 400 python -m pytest tests/test_gpu_e2e.py tests/test_gpu_server.py -x -q --timeout 200 > gpurun_out/r2_job17.log 2>&1
tail -8 gpurun_out/r2_job17.log
for NP in; do
timeout 300 python bench.py --batch 16 --series 1 --length 256 --steps 16 --warmup 4 $NP > gpurun_out/r2_bench_pack.json 2> gpurun_out/r2_bench_pack.err
python - <<PY
import json
d = json.loads(open("gpurun_out/r2_bench_pack.json").read().strip().splitlines()[-1])
print("pack='$NP' admit_total_ms", round(d["batch_admit_ms_total"], 2), "passes", d["packed_prefill_passes"], "ttft_p50", round(d["ttft_ms_p50"], 2), "tok/s", round(d["value"], 1), "prompt", d["config"]["prompt_tokens"])
PY
done
