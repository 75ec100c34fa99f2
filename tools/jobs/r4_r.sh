#!/bin/bash
R="${GRAFT_REPO_ROOT:-.}"; O=$R/gpurun_out; mkdir -p $O; cd $R
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 1200 python -m pytest tests/test_gpu_tp_multiprocess.py tests/test_gpu_tp_p2p.py tests/test_gpu_e2e.py -q -m gpu -x > $O/r4r_tests.log 2>&1; echo "tests rc=$?"; tail -3 $O/r4r_tests.log | cut -c1-300
for s in 0 1; do
  CHATTS_TP_SLABS=$s timeout 400 python tools/tp_shard_step.py --worlds 8 --batch 16 --weights fp8 --prefill-runs 1 --out $O/r4r_slabs$s.json > /dev/null 2> $O/r4r_slabs$s.err; echo "TP_SLABS=$s $(grep tp_shard_step $O/r4r_slabs$s.err | sed 's/.*batched_decode_ms_per_step": \([0-9.]*\).*/\1 ms per 16-wide step/')"
done
