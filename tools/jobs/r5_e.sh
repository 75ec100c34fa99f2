#!/bin/bash
# round 5, call E: ablations of the ring kernel (where do the cycles / the power go), one rank of TP = W with the ring kernel and the
# one-launch attention on / off, kernel trace of the W = 8 rank
set -x
mkdir -p gpurun_out/r5_e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
CHATTS_AMD_LIB=chatts_amd/lib/variants/libchatts_amd_probe.so timeout 300 python tools/ring_probe.py 798 > gpurun_out/r5_e/ring_ablate.txt 2>&1
cat gpurun_out/r5_e/ring_ablate.txt
CHATTS_ATTN_FOLD=0 timeout 900 python tools/tp_shard_step.py --worlds 1,2,4,8 --out gpurun_out/r5_e/tp_shard_step_nofold.json > gpurun_out/r5_e/tp_nofold.log 2>&1
tail -5 gpurun_out/r5_e/tp_nofold.log
timeout 600 python tools/tp_shard_step.py --worlds 8 --out gpurun_out/r5_e/tp_shard_step_fold.json > gpurun_out/r5_e/tp_fold.log 2>&1
tail -3 gpurun_out/r5_e/tp_fold.log
CHATTS_ATTN_FOLD=0 timeout 300 rocprofv3 --kernel-trace -d /tmp/kt8 -o p -- python tools/tp_shard_step.py --worlds 8 --steps 8 --out gpurun_out/r5_e/tp8_traced.json > gpurun_out/r5_e/tp8_trace.log 2>&1
python tools/prof_db.py $(find /tmp/kt8 -name "*.db" | head -1) > gpurun_out/r5_e/kt_tp8.txt
head -40 gpurun_out/r5_e/kt_tp8.txt
