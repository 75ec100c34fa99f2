#!/bin/bash
# round 5: the decode attention with the group size known to the compiler (ATTN_EXACT) and the few-row split-K epilogue that takes
# three column groups per round trip (EPI_NORM_Q_GROUPS): parity tests, then interleaved A/B on the headline and on config 5
R="${GRAFT_REPO_ROOT:-.}"; O=$R/gpurun_out/r5_k; mkdir -p $O; cd $R
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_e2e.py tests/test_gpu_parity_real_size.py -q -m gpu -x -k "attn or attention or decode or post_norm or batch or config5 or bench_prompt" > $O/pytest_subset.txt 2>&1; tail -3 $O/pytest_subset.txt
line() { python - "$1" <<'PY'
import json, sys
try:
    r = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print("   %-44s %8.1f tok/s  %.3f ms/step  ttft %.2f" % (sys.argv[1].split("/")[-1], r["value"], r["ms_per_step"], r.get("ttft_ms_p50") or -1))
except Exception as e:
    print(sys.argv[1], "ERR", e)
PY
}
echo "== headline (bf16, batch 1): ATTN_EXACT 0 / 1, interleaved"
for rep in 1 2; do for ex in 0 1; do
  CHATTS_ATTN_EXACT=$ex timeout 300 python bench.py --steps 48 --warmup 8 --no-cpu-baseline --ttft-runs 1 > $O/head_exact${ex}_$rep.json 2> $O/err.txt; line $O/head_exact${ex}_$rep.json
done; done
echo "== config 5 (fp8 weights, 16 x (8 x 1024)): (ATTN_EXACT, EPI_NORM_Q_GROUPS)"
C5="--batch 16 --weights fp8 --series 8 --length 1024 --steps 24 --warmup 6 --no-cpu-baseline"
for rep in 1 2; do for arm in "0 1" "1 1" "0 3" "1 3"; do set -- $arm
  CHATTS_ATTN_EXACT=$1 CHATTS_EPI_NORM_Q_GROUPS=$2 timeout 400 python bench.py $C5 > $O/cfg5_exact$1_g$2_$rep.json 2> $O/err.txt; line $O/cfg5_exact$1_g$2_$rep.json
done; done
echo "== config 5, shipped options, slots per sequence"
for ns in 8 16 24 32; do
  CHATTS_BATCH_NSPLITS=$ns timeout 400 python bench.py $C5 > $O/cfg5_ns$ns.json 2> $O/err.txt; line $O/cfg5_ns$ns.json
done
cd /tmp && export TMPDIR=/tmp; rm -rf /tmp/kt5
timeout 400 rocprofv3 --kernel-trace -d /tmp/kt5 -o p -- python $R/bench.py --batch 16 --weights fp8 --series 8 --length 1024 --steps 8 --warmup 2 --no-cpu-baseline > /tmp/kt5.log 2>&1
(echo "## rocprofv3 --kernel-trace -- python bench.py --batch 16 --weights fp8 --series 8 --length 1024 --steps 8 --warmup 2 --no-cpu-baseline   (MI355X, round 5)"; python $R/tools/prof_db.py $(find /tmp/kt5 -name "*.db" | head -1) | grep -v fill_hash | head -50) > $O/r5_cfg5_kernel_trace.txt
grep -n "attn_decode\|norm_q" $O/r5_cfg5_kernel_trace.txt | cut -c1-150
