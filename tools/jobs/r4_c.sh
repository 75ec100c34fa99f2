#!/bin/bash
# round 4, call C: one process per rank (8 processes, one GPU) against the oracle; host-counted epochs; per-rank step again
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out; O=gpurun_out; R=$PWD
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 1200 python -m pytest tests/test_gpu_tp_multiprocess.py tests/test_gpu_tp_p2p.py -m gpu -q -s -x > $O/r4c_tests.log 2>&1; echo "tests rc=$?"; tail -30 $O/r4c_tests.log | cut -c1-600
timeout 600 python tools/tp_shard_step.py --worlds 2,4,8 --out $O/r4c_shard.json > /dev/null 2> $O/r4c_shard.err; grep tp_shard_step $O/r4c_shard.err
export CHATTS_FORCE_DEVICE=0 CHATTS_DIST_BACKEND=gloo CHATTS_TP_FUSE_BLOCKS=48 OMP_NUM_THREADS=16
timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 --master-port 29541 tools/tp_parity_worker.py --flow config4 --out $O/r4_tp8_parity_config4.json > $O/r4c_cfg4.log 2>&1; echo "config4 rc=$?"; tail -3 $O/r4c_cfg4.log | cut -c1-1500
timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 4 --master-addr 127.0.0.1 --master-port 29543 tools/tp_parity_worker.py --flow headline --out $O/r4_tp4_parity_headline.json > $O/r4c_tp4.log 2>&1; echo "tp4 headline rc=$?"; tail -2 $O/r4c_tp4.log | cut -c1-1500
