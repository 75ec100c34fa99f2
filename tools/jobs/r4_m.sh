#!/bin/bash
# round 4, call M: the TP bit-identity test again, the qkv-slab fold (tests + config 5 A/B), per-rank batched step
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out; O=gpurun_out
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 900 python -m pytest tests/test_gpu_tp_p2p.py tests/test_gpu_e2e.py tests/test_gpu_parity_real_size.py -q -x -m gpu > $O/r4m_tests.log 2>&1; echo "tests rc=$?"; tail -4 $O/r4m_tests.log | cut -c1-300
for f in 0 1; do
  CHATTS_QKV_FOLD=$f timeout 400 python bench.py --batch 16 --weights fp8 --series 8 --length 1024 --steps 24 --warmup 6 --no-cpu-baseline > $O/r4m_cfg5_fold$f.json 2> $O/r4m_cfg5_fold$f.err
  python -c "
import json; d=json.loads(open('$O/r4m_cfg5_fold$f.json').read().strip().splitlines()[-1]); print('cfg5 QKV_FOLD=$f', round(d['value'],1), 'tok/s', round(d['ms_per_step'],3), 'ms/step parity', d.get('parity_checked'), 'ttft', round(d['ttft_ms_p50'],1))"
done
timeout 400 python tools/tp_shard_step.py --worlds 8 --batch 16 --weights fp8 --prefill-runs 1 --out $O/r4m_tp8_cfg5.json > /dev/null 2> $O/r4m_tp8_cfg5.err; grep tp_shard_step $O/r4m_tp8_cfg5.err | cut -c1-300
