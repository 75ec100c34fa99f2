#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out
timeout 300 python -m pytest tests/test_gpu_kernels.py -x -q --timeout 100 -k "stream_k or gemm_dma" > gpurun_out/r2_job14.log 2>&1
tail -12 gpurun_out/r2_job14.log
for SKM in 0 -1; do
if [ $SKM = -1 ]; then unset CHATTS_GEMM_STREAMK; else export CHATTS_GEMM_STREAMK=$SKM; fi
timeout 300 python bench.py --steps 8 --warmup 2 --no-cpu-baseline --ttft-runs 5 > gpurun_out/r2_bench_sk.json 2> gpurun_out/r2_bench_sk.err
python - <<PY
import json
d = json.loads(open("gpurun_out/r2_bench_sk.json").read().strip().splitlines()[-1])
print("STREAMK=$SKM ttft", round(d["ttft_ms_p50"], 3), "tok/s", round(d["value"], 1), "parity", d["parity_checked"])
PY
done
