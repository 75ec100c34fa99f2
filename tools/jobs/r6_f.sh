#!/bin/bash
# round 6, job f: tiled prefill weights in the decoder - tests, bench line
R="${GRAFT_REPO_ROOT:-.}"; O=$R/gpurun_out/r6_f; mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_gpu_kernels.py -m gpu -x -q -k "tiled" > $O/pytest_tiled.txt 2>&1
tail -3 $O/pytest_tiled.txt
timeout 900 python -m pytest tests/test_gpu_e2e.py -m gpu -x -q > $O/pytest_e2e.txt 2>&1
tail -3 $O/pytest_e2e.txt
timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_tiled.json 2> $O/bench_tiled.err
CHATTS_TILED_WEIGHTS=0 timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_rowmajor.json 2> $O/bench_rowmajor.err
python - <<'PY'
import json
for n in ("tiled", "rowmajor"):
    try:
        d = json.loads(open(f"gpurun_out/r6_f/bench_{n}.json").read().strip().splitlines()[-1])
        print(n, "tok/s", round(d["value"], 2), "ttft", round(d["ttft_ms_p50"], 2), "parity", d["parity_checked"], "gate_up us", round(d["prefill_roofline"]["avg_us"], 1),
              "tiled GB", d["config"]["weight_bytes_tiled_prefill_copies"] / 1e9)
    except Exception as e:
        print(n, "failed", e)
PY
tail -3 $O/bench_tiled.err
