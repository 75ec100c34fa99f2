#!/bin/bash
# round 3, job 1: real-size parity tests for configs 4 / 5, their committed full-depth oracle runs, their bench lines (+ the cfg 5
# kernel trace), and the self-launching TP paths on one device (two processes time-slicing the GPU: correctness only).
R="${GRAFT_REPO_ROOT:-.}"
cd "$R"; mkdir -p gpurun_out
export TMPDIR=/tmp
( timeout 900 python -m pytest tests/test_gpu_parity_real_size.py -x -q --timeout 600 2>&1 | tail -5 ) > gpurun_out/r3_job1_pytest.log 2>&1
tail -3 gpurun_out/r3_job1_pytest.log
timeout 900 python tools/parity_full_depth.py --series 30 --lengths mixed --out gpurun_out/r3_parity_14b_30xmixed_bf16_b1_full.json > gpurun_out/r3_parity_cfg4.log 2>&1
echo "cfg4 parity rc=$?"; tail -c 600 gpurun_out/r3_parity_cfg4.log
timeout 1200 python tools/parity_full_depth.py --series 8 --length 1024 --batch 16 --weights fp8 --oracle-slots 0,7,15 --new 6 \
    --out gpurun_out/r3_parity_14b_8x1024_fp8_b16_full.json > gpurun_out/r3_parity_cfg5.log 2>&1
echo "cfg5 parity rc=$?"; tail -c 600 gpurun_out/r3_parity_cfg5.log
cp gpurun_out/r3_parity_14b_*_full.json profiles/ 2>/dev/null
timeout 600 python bench.py --series 30 --lengths mixed --steps 32 --warmup 8 --no-cpu-baseline > gpurun_out/r3_bench_cfg4_30xmixed.json 2> gpurun_out/r3_bench_cfg4.err
echo "cfg4 bench rc=$?"; tail -c 300 gpurun_out/r3_bench_cfg4.err
timeout 600 python bench.py --batch 16 --weights fp8 --series 8 --length 1024 --steps 32 --warmup 8 > gpurun_out/r3_bench_cfg5_fp8_8x1024_b16.json 2> gpurun_out/r3_bench_cfg5.err
echo "cfg5 bench rc=$?"; tail -c 300 gpurun_out/r3_bench_cfg5.err
cd /tmp; rm -rf /tmp/kt5
timeout 400 rocprofv3 --kernel-trace -d /tmp/kt5 -o p -- python $R/bench.py --batch 16 --weights fp8 --series 8 --length 1024 --steps 8 --warmup 2 > /tmp/kt5.log 2>&1
db=$(find /tmp/kt5 -name "*.db" | head -1)
(echo "## rocprofv3 --kernel-trace -- python bench.py --batch 16 --weights fp8 --series 8 --length 1024 --steps 8 --warmup 2   (MI355X, round 3, before the cfg-5 work)"; python $R/tools/prof_db.py $db) > $R/gpurun_out/r3_cfg5_kernel_trace_before.txt 2>&1
cd "$R"
# TP=2, two processes on the one GPU, launched by bench.py itself
export HSA_ENABLE_IPC_MODE_LEGACY=0
CHATTS_FORCE_DEVICE=0 CHATTS_DIST_BACKEND=gloo timeout 600 python bench.py --gpus 2 --no-cpu-baseline --steps 16 --warmup 4 --ttft-runs 2 \
    > gpurun_out/r3_tp2_self_launch_single_device.json 2> gpurun_out/r3_tp2_self_launch.err
echo "tp2 self-launch rc=$?"; tail -c 400 gpurun_out/r3_tp2_self_launch_single_device.json | head -c 400; tail -3 gpurun_out/r3_tp2_self_launch.err
CHATTS_FORCE_DEVICE=0 CHATTS_DIST_BACKEND=gloo timeout 300 python tools/llm_tp_spawn_check.py > gpurun_out/r3_llm_tp2_spawn_single_device.json 2> gpurun_out/r3_llm_tp2_spawn.err
echo "llm spawn rc=$?"; cat gpurun_out/r3_llm_tp2_spawn_single_device.json | tail -c 600; tail -3 gpurun_out/r3_llm_tp2_spawn.err
python - <<PY
import json
for f in ("r3_bench_cfg4_30xmixed.json", "r3_bench_cfg5_fp8_8x1024_b16.json"):
    try:
        d = json.loads(open("gpurun_out/" + f).read().strip().splitlines()[-1])
        print(f, d["value"], d["ms_per_step"], d.get("ttft_ms_p50"), d["parity_checked"], d["parity"], d["config"].get("prompt_tokens"))
    except Exception as e:
        print(f, "unreadable", e)
PY
head -30 gpurun_out/r3_cfg5_kernel_trace_before.txt | cut -c1-180
