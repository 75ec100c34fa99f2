#!/bin/bash
# round 4, call E: decode attention refactor (two-kernel form untouched, workgroup form opt-in) + its A/B; the 8-process worker with logs
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out; O=gpurun_out; R=$PWD
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 900 python -m pytest tests/test_gpu_kernels.py -m gpu -q -x -k "attn or attention" > $O/r4e_tests.log 2>&1; echo "attention tests rc=$?"; tail -4 $O/r4e_tests.log | cut -c1-300
( export CHATTS_FORCE_DEVICE=0 CHATTS_DIST_BACKEND=gloo CHATTS_TP_FUSE_BLOCKS=48 OMP_NUM_THREADS=16
  timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 --master-port 29551 tools/tp_parity_worker.py --flow headline --out $O/r4_tp8_parity_headline.json > $O/r4e_tp8.log 2>&1; echo "tp8 headline rc=$?"; grep -h "tp_parity_worker rank" -A25 $O/r4e_tp8.log | head -60 | cut -c1-300; tail -2 $O/r4e_tp8.log | cut -c1-1200 )
for m in 0 1; do
  CHATTS_ATTN_WG=$m timeout 400 python bench.py --batch 16 --weights fp8 --series 8 --length 1024 --steps 24 --warmup 6 --no-cpu-baseline > $O/r4e_cfg5_wg$m.json 2> $O/r4e_cfg5_wg$m.err; echo "cfg5 wg=$m rc=$?"
  python -c "
import json; d=json.loads(open('$O/r4e_cfg5_wg$m.json').read().strip().splitlines()[-1]); print('cfg5 ATTN_WG=$m', round(d['value'],1), 'tok/s', round(d['ms_per_step'],3), 'ms/step parity', d.get('parity_checked'))"
done
for m in 0 2; do
  CHATTS_ATTN_WG=$m timeout 300 python tools/tp_shard_step.py --worlds 1,8 --prefill-runs 1 --out $O/r4e_shard_wg$m.json > /dev/null 2> $O/r4e_shard_wg$m.err; echo "shard ATTN_WG=$m"; grep tp_shard_step $O/r4e_shard_wg$m.err | cut -c1-260
done
