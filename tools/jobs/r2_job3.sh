#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_tp_p2p.py -x -q > gpurun_out/r2_tp_tests.log 2>&1
tail -12 gpurun_out/r2_tp_tests.log
timeout 900 python -m pytest tests/test_gpu_kernels.py -x -q -k "ts_" > gpurun_out/r2_ts_tests.log 2>&1
tail -12 gpurun_out/r2_ts_tests.log
timeout 300 python bench.py --steps 16 --warmup 4 --no-cpu-baseline --ttft-runs 3 > gpurun_out/r2_bench_b.json 2> gpurun_out/r2_bench_b.err
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r2_bench_b.json").read().strip().splitlines()[-1])
print("tok/s", d["value"], "ttft", d["ttft_ms_p50"], "ts_enc_ms", d["ts_encode_ms_p50"])
print(json.dumps(d.get("ts_encoder_roofline")))
PY
(time timeout 2400 python -m pytest tests -x -q -m gpu) > gpurun_out/r2_gpu_suite.log 2>&1
tail -8 gpurun_out/r2_gpu_suite.log
