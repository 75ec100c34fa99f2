#!/bin/bash
# round 5, call I: bulk all-reduce with all of a thread's positions in flight per pass - sweep, TP tests, the loop-back rank steps
set -x
O=gpurun_out/r5_i; mkdir -p $O
cd /tmp && export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0 && cd $GRAFT_REPO_ROOT
timeout 300 python tools/tp_bulk_sweep.py 8 798 > $O/tp_bulk_sweep_w8.txt 2>&1; cat $O/tp_bulk_sweep_w8.txt
timeout 300 python tools/tp_bulk_sweep.py 2 798 > $O/tp_bulk_sweep_w2.txt 2>&1; cat $O/tp_bulk_sweep_w2.txt
timeout 600 python -m pytest tests/test_gpu_tp_p2p.py tests/test_gpu_tp_multiprocess.py -q -m gpu > $O/pytest_tp.txt 2>&1; tail -5 $O/pytest_tp.txt
timeout 600 python tools/tp_shard_step.py --worlds 2,8 --out $O/tp_shard_step.json > $O/tp.log 2>&1; tail -3 $O/tp.log | cut -c1-400
