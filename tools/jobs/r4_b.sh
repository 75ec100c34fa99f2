#!/bin/bash
# round 4, call B: the remaining TP tests (config 5 on TP=8 shards, the exchange inside the GEMV launch vs the stand-alone kernels), the
# K-split GEMV of the shard shapes, and the per-rank step with it (auto, forced 2, forced 5)
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out; O=gpurun_out; R=$PWD
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 900 python -m pytest tests/test_gpu_tp_p2p.py "tests/test_gpu_kernels.py::test_gemv_ksplit_at_tensor_parallel_shard_shapes" "tests/test_gpu_tp_shards.py::test_config5_fp8_batch16_on_tp8_shards_vs_oracle" "tests/test_gpu_tp_shards.py::test_headline_prompt_on_tp_shards_vs_oracle" -m gpu -q -s > $O/r4b_tests.log 2>&1; echo "tests rc=$?"; tail -12 $O/r4b_tests.log
timeout 600 python tools/tp_shard_step.py --worlds 2,4,8 --out $O/r4b_shard_auto.json > /dev/null 2> $O/r4b_shard_auto.err; grep tp_shard_step $O/r4b_shard_auto.err
CHATTS_GEMV_KS=2 timeout 600 python tools/tp_shard_step.py --worlds 8 --prefill-runs 1 --out $O/r4b_shard_ks2.json > /dev/null 2> $O/r4b_shard_ks2.err; echo ks2; grep tp_shard_step $O/r4b_shard_ks2.err
CHATTS_GEMV_KS=5 timeout 600 python tools/tp_shard_step.py --worlds 4,8 --prefill-runs 1 --out $O/r4b_shard_ks5.json > /dev/null 2> $O/r4b_shard_ks5.err; echo ks5; grep tp_shard_step $O/r4b_shard_ks5.err
cd /tmp && export TMPDIR=/tmp; rm -rf /tmp/kt8
timeout 600 rocprofv3 --kernel-trace -d /tmp/kt8 -o p -- python $R/tools/tp_shard_step.py --worlds 8 --steps 8 --warmup 2 --prefill-runs 1 --out $R/$O/r4_tp8_traced.json > /tmp/kt8.log 2>&1; echo "rocprof rc=$?"
db=$(find /tmp/kt8 -name "*.db" | head -1)
(echo "## rocprofv3 --kernel-trace -- python tools/tp_shard_step.py --worlds 8 --steps 8 --warmup 2 --prefill-runs 1   (ONE rank of TP=8, loop-back exchange, K-split GEMV auto, MI355X, round 4)"; python $R/tools/prof_db.py $db) > $R/$O/r4b_tp8_shard_kernel_trace.txt
cd $R; head -16 $O/r4b_tp8_shard_kernel_trace.txt
