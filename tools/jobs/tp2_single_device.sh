#!/bin/bash
# Two PROCESSES (one per TP rank) on the one GPU of a gpurun box: the ranks exchange hipIpc handles over gloo and map each other's
# exchange buffers, then run bench.py's TP=2 path (whole-step hipGraph incl. chatts_allreduce / chatts_tp_argmax).
# Correctness evidence for the cross-process exchange; the xGMI latency itself needs a multi-GPU node.
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out
export HSA_ENABLE_IPC_MODE_LEGACY=0 CHATTS_FORCE_DEVICE=0 CHATTS_DIST_BACKEND=gloo
ARGS="${TP2_ARGS:---steps 16 --warmup 4 --no-cpu-baseline --ttft-runs 2}"
timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 \
    bench.py --gpus 2 $ARGS > gpurun_out/r2_tp2_single_device.json 2> gpurun_out/r2_tp2_single_device.err
echo "tp2 rc=$?"; tail -c 1500 gpurun_out/r2_tp2_single_device.json; tail -5 gpurun_out/r2_tp2_single_device.err
