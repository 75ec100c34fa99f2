#!/bin/bash
# config 5: FETCH_SIZE pass (its own run: --pmc with --kernel-trace only) + the bench line with the round-3 fields; then the default bench line
R="${GRAFT_REPO_ROOT:-.}"; cd /tmp; export TMPDIR=/tmp; mkdir -p $R/gpurun_out; rm -rf /tmp/fs5
timeout 500 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d /tmp/fs5 -o p -- python $R/bench.py --batch 16 --weights fp8 --series 8 --length 1024 --steps 4 --warmup 2 --no-cpu-baseline > /tmp/fs5.log 2>&1
db=$(find /tmp/fs5 -name "*.db" | head -1)
(echo "## rocprofv3 --kernel-trace --pmc FETCH_SIZE -- python bench.py --batch 16 --weights fp8 --series 8 --length 1024 --steps 4 --warmup 2 --no-cpu-baseline   (MI355X, round 3)"; python $R/tools/prof_db.py $db) > $R/gpurun_out/r3_cfg5_pmc_fetch_size.txt 2>&1
grep -n "FETCH_SIZE" $R/gpurun_out/r3_cfg5_pmc_fetch_size.txt | grep "gemm_stream\|attn_decode" | head -8 | cut -c1-220
cd $R
timeout 400 python bench.py --batch 16 --weights fp8 --series 8 --length 1024 --steps 24 --warmup 6 --no-cpu-baseline > gpurun_out/r3_bench_cfg5.json 2> gpurun_out/r3_bench_cfg5.err
python - <<PY
import json
d=json.loads(open("gpurun_out/r3_bench_cfg5.json").read().strip().splitlines()[-1])
print("cfg5", d["value"], d["ms_per_step"], d["parity_checked"], d["decode_hbm_frac_of_8TBs_incl_kv"], json.dumps(d["ts_encoder_roofline"])[:400])
PY
timeout 600 python bench.py --steps 32 --warmup 8 > gpurun_out/r3_bench_n1.json 2> gpurun_out/r3_bench_n1.err
python - <<PY
import json
d=json.loads(open("gpurun_out/r3_bench_n1.json").read().strip().splitlines()[-1])
print("n1", d["value"], d["ms_per_step"], d.get("ttft_ms_p50"), d["parity_checked"], d["roofline"]["frac"], {k: (v if not isinstance(v, str) else v[:80]) for k, v in d["cpu_baseline"].items() if k in ("value","cores","kind","ts_encode_ms","ts_encode_patches","wall_s")})
PY
