#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out
timeout 300 python -m pytest tests/test_gpu_kernels.py -x -q --timeout 100 -k "stream or ts_ or gemm_dma_parity or plane" > gpurun_out/r2_job15.log 2>&1
tail -8 gpurun_out/r2_job15.log
timeout 200 python - <<'PY'
import sys, os, torch
sys.path.insert(0, os.getcwd())
sys.argv = ["x", "128"]
import tools.ts_gemm_sweep as t
for K in (5120, 320):
    for env in ({}, {"CHATTS_GEMM_STREAM_MB": 0}, {"CHATTS_GEMM_SK": 4}, {"CHATTS_GEMM_SK": 8}, {"CHATTS_GEMM_SK": 3}):
        print("P=128 K=%d" % K, env, round(t.run(128, K, 5120, env, nbuf=8, reps=3), 2), "us")
for P in (64, 32):
    print("P=%d K=5120" % P, round(t.run(P, 5120, 5120, {}, nbuf=8, reps=3), 2), "us", "dma:", round(t.run(P, 5120, 5120, {"CHATTS_GEMM_STREAM_MB": 0}, nbuf=8, reps=3), 2))
PY
timeout 200 python bench.py --steps 8 --warmup 2 --no-cpu-baseline --ttft-runs 3 > gpurun_out/r2_bench_ts.json 2> gpurun_out/r2_bench_ts.err
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r2_bench_ts.json").read().strip().splitlines()[-1])
print("ttft", round(d["ttft_ms_p50"], 3), "tok/s", round(d["value"], 1), "parity", d["parity_checked"], json.dumps(d["ts_encoder_roofline"]))
PY
