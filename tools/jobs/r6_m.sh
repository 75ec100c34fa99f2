#!/bin/bash
# round 6, job m: bench line of the opt-in f16q prefill mode; then the whole GPU suite + smoke
R="${GRAFT_REPO_ROOT:-.}"; O=$R/gpurun_out/r6_m; mkdir -p $O
cd $R
timeout 600 python bench.py --precision f16q --steps 20 --warmup 5 --no-cpu-baseline > $O/r6_bench_f16q.json 2> $O/bench_f16q.err
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r6_m/r6_bench_f16q.json").read().strip().splitlines()[-1])
print("f16q: tok/s", round(d["value"], 2), "ttft", round(d["ttft_ms_p50"], 2), "parity", d["parity_checked"], d["parity"].get("source"), "f16q GB", d["config"]["weight_bytes_f16q_copies"] / 1e9)
PY
timeout 2400 python -m pytest tests/ -x -q -m gpu > $O/pytest_gpu.txt 2>&1
tail -6 $O/pytest_gpu.txt | cut -c1-300
timeout 600 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.txt 2>&1
tail -2 $O/smoke.txt | cut -c1-300
