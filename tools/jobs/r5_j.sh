#!/bin/bash
# round 5, call J: bulk all-reduce release form A/B (drain only vs __threadfence_system), TP tests with the drain-only form,
# fused norm epilogue (registers) bitwise test + A/B in the bench
set -x
O=gpurun_out/r5_j; mkdir -p $O
cd /tmp && export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0 && cd $GRAFT_REPO_ROOT
timeout 300 python tools/tp_bulk_sweep.py 8 798 0 > $O/tp_bulk_sweep_w8_light.txt 2>&1; cat $O/tp_bulk_sweep_w8_light.txt
timeout 300 python tools/tp_bulk_sweep.py 8 798 1 > $O/tp_bulk_sweep_w8_fence.txt 2>&1; cat $O/tp_bulk_sweep_w8_fence.txt
timeout 300 python tools/tp_bulk_sweep.py 2 798 0 > $O/tp_bulk_sweep_w2_light.txt 2>&1; tail -7 $O/tp_bulk_sweep_w2_light.txt
timeout 600 python -m pytest tests/test_gpu_tp_p2p.py tests/test_gpu_tp_multiprocess.py -q -m gpu > $O/pytest_tp.txt 2>&1; tail -5 $O/pytest_tp.txt
timeout 400 python -m pytest tests/test_gpu_kernels.py -q -m gpu -k "post_norm or attention_decode or attention_parity" > $O/pytest_postnorm.txt 2>&1; tail -5 $O/pytest_postnorm.txt
for reg in 1 0; do
  CHATTS_EPI_NORM_REG=$reg timeout 300 python bench.py --steps 8 --warmup 2 --no-cpu-baseline --ttft-runs 5 > $O/bench_reg$reg.txt 2>&1
  python - $O/bench_reg$reg.txt <<'PY'
import json, sys
r = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(sys.argv[1], "ttft", r["ttft_ms_p50"], "tok/s", r["value"], "parity", r["parity_checked"])
PY
done
CHATTS_EPI_NORM_REG=1 timeout 300 python bench.py --steps 8 --warmup 2 --no-cpu-baseline --ttft-runs 5 2>/dev/null | tail -1 | python -c "import json,sys; r=json.loads(sys.stdin.read()); print('again reg1 ttft', r['ttft_ms_p50'])"
timeout 600 python tools/tp_shard_step.py --worlds 8 --out $O/tp_shard_step_light.json > $O/tp.log 2>&1; grep "W=8" $O/tp.log | cut -c1-300
CHATTS_TP_BULK_FENCE=1 timeout 600 python tools/tp_shard_step.py --worlds 8 --out $O/tp_shard_step_fence.json > $O/tp2.log 2>&1; grep "W=8" $O/tp2.log | cut -c1-300
