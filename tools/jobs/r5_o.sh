#!/bin/bash
# round 5: decode attention, exact form, probabilities handed to P.V through LDS instead of v_readlane + select: parity tests, A/B against
# the run-time form (ATTN_EXACT=0, unchanged code: the common baseline with profiles/r5_attn_exact_normq_ab.txt), then - the sources of the
# batched digest changed - the config-5 FETCH_SIZE pass and the bench lines that read it
R="${GRAFT_REPO_ROOT:-.}"; O=$R/gpurun_out/r5_o; mkdir -p $O; cd $R
export HSA_ENABLE_IPC_MODE_LEGACY=0 PMC_TRAFFIC_OUT=$R/profiles/pmc_traffic.json
timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_e2e.py tests/test_gpu_parity_real_size.py tests/test_gpu_tp_shards.py -q -m gpu -x -k "attn or attention or decode or batch or config5 or config4 or bench_prompt or headline" > $O/pytest_subset.txt 2>&1; tail -2 $O/pytest_subset.txt
line() { python - "$1" <<'PY'
import json, sys
try:
    r = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print("   %-44s %8.1f tok/s  %.3f ms/step  ttft %.2f  parity %s traffic %s" % (sys.argv[1].split("/")[-1], r["value"], r["ms_per_step"], r.get("ttft_ms_p50") or -1, r.get("parity_checked"), (r.get("roofline") or {}).get("traffic")))
except Exception as e:
    print(sys.argv[1], "ERR", e)
PY
}
C5="--batch 16 --weights fp8 --series 8 --length 1024 --steps 24 --warmup 6 --no-cpu-baseline"
for rep in 1 2; do for ex in 0 1; do
  CHATTS_ATTN_EXACT=$ex timeout 300 python bench.py --steps 48 --warmup 8 --no-cpu-baseline --ttft-runs 1 > $O/head_exact${ex}_$rep.json 2> $O/err.txt; line $O/head_exact${ex}_$rep.json
  CHATTS_ATTN_EXACT=$ex timeout 400 python bench.py $C5 > $O/cfg5_exact${ex}_$rep.json 2> $O/err.txt; line $O/cfg5_exact${ex}_$rep.json
done; done
for ex in 0 1; do
  CHATTS_ATTN_EXACT=$ex timeout 400 python bench.py --series 30 --lengths mixed --steps 32 --warmup 8 --no-cpu-baseline --ttft-runs 1 > $O/cfg4_exact${ex}.json 2> $O/err.txt; line $O/cfg4_exact${ex}.json
done
cd /tmp && export TMPDIR=/tmp; rm -rf /tmp/fs5
timeout 500 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d /tmp/fs5 -o p -- python $R/bench.py --batch 16 --weights fp8 --series 8 --length 1024 --steps 4 --warmup 2 --no-cpu-baseline > /tmp/fs5.log 2>&1
db=$(find /tmp/fs5 -name "*.db" | head -1)
(echo "## rocprofv3 --kernel-trace --pmc FETCH_SIZE -- python bench.py $C5   (steps 4, warmup 2; MI355X, round 5, final code)"; python $R/tools/prof_db.py $db | grep -v fill_hash | head -60) > $O/r5_cfg5_pmc_fetch_size.txt
( cd $R && python tools/pmc_traffic.py batched $db "rocprofv3 --kernel-trace --pmc FETCH_SIZE -- python bench.py --batch 16 --weights fp8 --series 8 --length 1024 --steps 4 --warmup 2 --no-cpu-baseline" profiles/r5_cfg5_pmc_fetch_size.txt | cut -c1-300 )
cp $R/profiles/pmc_traffic.json $O/pmc_traffic.json
cd $R
timeout 400 python bench.py $C5 > $O/r5_bench_cfg5_fp8_8x1024_b16.json 2> $O/err.txt; line $O/r5_bench_cfg5_fp8_8x1024_b16.json
timeout 600 python bench.py --steps 32 --warmup 8 > $O/r5_bench_n1.json 2> $O/err.txt; line $O/r5_bench_n1.json
timeout 400 python bench.py --series 30 --lengths mixed --steps 32 --warmup 8 --no-cpu-baseline > $O/r5_bench_cfg4_30xmixed.json 2> $O/err.txt; line $O/r5_bench_cfg4_30xmixed.json
timeout 300 python bench.py --model chatts-8b --series 1 --length 256 --steps 32 --warmup 8 --no-cpu-baseline > $O/r5_bench_8b_cfg2.json 2> $O/err.txt; line $O/r5_bench_8b_cfg2.json
