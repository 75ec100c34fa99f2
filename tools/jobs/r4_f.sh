#!/bin/bash
# round 4, call F: where does the 8-process run of the bulk all-reduce stall?  W = 2 / 4 / 8, fewer workgroups, bulk off
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out; O=gpurun_out; R=$PWD
export HSA_ENABLE_IPC_MODE_LEGACY=0 CHATTS_FORCE_DEVICE=0 CHATTS_DIST_BACKEND=gloo CHATTS_TP_FUSE_BLOCKS=48 OMP_NUM_THREADS=16
run() { # name world extra-env...
  name=$1; W=$2; shift 2
  env "$@" timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node $W --master-addr 127.0.0.1 --master-port $((29560 + RANDOM % 200)) tools/tp_parity_worker.py --flow headline --out $O/r4f_$name.json > $O/r4f_$name.log 2>&1
  echo "$name rc=$? $(grep -c 'timed out' $O/r4f_$name.log) timeouts; $(tail -1 $O/r4f_$name.log | cut -c1-300)"
}
run tp2 2 A=1
run tp4 4 A=1
run tp8_blocks16 8 CHATTS_TP_BULK_BLOCKS=16
run tp8_blocks4 8 CHATTS_TP_BULK_BLOCKS=4
run tp8_bulk_off 8 CHATTS_TP_BULK=0
run tp8_default 8 A=1
