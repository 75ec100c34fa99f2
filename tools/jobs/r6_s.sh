#!/bin/bash
# round 6, job s: decode GEMV with the first weight chunks requested before the prologue - tests, then interleaved bench A/B against the previous kernel (variant library)
R="${GRAFT_REPO_ROOT:-.}"; O=$R/gpurun_out/r6_s; mkdir -p $O
cd $R
timeout 1200 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_e2e.py -m gpu -x -q -k "gemv or generate_matches or graph_replay" > $O/pytest.txt 2>&1; tail -2 $O/pytest.txt
for rep in 1 2 3; do
  for arm in new old; do
    if [ $arm = old ]; then export CHATTS_AMD_LIB=$R/chatts_amd/lib/variants/libchatts_amd_oldgemv.so; else unset CHATTS_AMD_LIB; fi
    timeout 300 python bench.py --steps 64 --warmup 8 --no-cpu-baseline --ttft-runs 2 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$arm', 'tok/s', round(d['value'],2), 'ms/step', round(d['ms_per_step'],4), 'dominant us', round(d['roofline']['avg_us'],2), d['parity_checked'])" | tee -a $O/ab.txt
  done
done
unset CHATTS_AMD_LIB
