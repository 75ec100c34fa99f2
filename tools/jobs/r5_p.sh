#!/bin/bash
# round 5, last GPU job: decode attention form chosen by the host (batch 1: group compiled in, v_readlane P.V; batched: + probabilities
# through LDS) - the bit-identity tests of the three forms, the config-5 FETCH_SIZE pass on the final sources, the bench lines that read it
R="${GRAFT_REPO_ROOT:-.}"; O=$R/gpurun_out/r5_p; mkdir -p $O; cd $R
export HSA_ENABLE_IPC_MODE_LEGACY=0 PMC_TRAFFIC_OUT=$R/profiles/pmc_traffic.json
timeout 300 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_e2e.py -q -m gpu -x -k "attention_decode or batched or continuous" > $O/pytest_subset.txt 2>&1; tail -2 $O/pytest_subset.txt
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
cd /tmp && export TMPDIR=/tmp; rm -rf /tmp/fs5
timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d /tmp/fs5 -o p -- python $R/bench.py --batch 16 --weights fp8 --series 8 --length 1024 --steps 4 --warmup 2 --no-cpu-baseline > /tmp/fs5.log 2>&1
db=$(find /tmp/fs5 -name "*.db" | head -1)
(echo "## rocprofv3 --kernel-trace --pmc FETCH_SIZE -- python bench.py --batch 16 --weights fp8 --series 8 --length 1024 --steps 4 --warmup 2 --no-cpu-baseline   (MI355X, round 5, final code)"; python $R/tools/prof_db.py $db | grep -v fill_hash | head -60) > $O/r5_cfg5_pmc_fetch_size.txt
( cd $R && python tools/pmc_traffic.py batched $db "rocprofv3 --kernel-trace --pmc FETCH_SIZE -- python bench.py --batch 16 --weights fp8 --series 8 --length 1024 --steps 4 --warmup 2 --no-cpu-baseline" profiles/r5_cfg5_pmc_fetch_size.txt | cut -c1-200 )
cp $R/profiles/pmc_traffic.json $O/pmc_traffic.json
cd $R
timeout 300 python bench.py $C5 > $O/r5_bench_cfg5_fp8_8x1024_b16.json 2> $O/err.txt; line $O/r5_bench_cfg5_fp8_8x1024_b16.json
timeout 300 python bench.py --steps 32 --warmup 8 --no-cpu-baseline --ttft-runs 2 > $O/head.json 2> $O/err.txt; line $O/head.json
grep -n "attn_decode_kernel" $O/r5_cfg5_pmc_fetch_size.txt | cut -c1-150
