#!/bin/bash
# round-3 re-check of the tensor-parallel paths on the one GPU of a gpurun box (two ranks time-slice it: correctness only):
# (1) python bench.py --gpus 2 launching its own ranks, (2) LLM(tensor_parallel_size=2) spawning its follower, (3) the rope test
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out
export HSA_ENABLE_IPC_MODE_LEGACY=0 CHATTS_FORCE_DEVICE=0 CHATTS_DIST_BACKEND=gloo
timeout 600 python bench.py --gpus 2 --steps 16 --warmup 4 --no-cpu-baseline --ttft-runs 2 > gpurun_out/r3_tp2_self_launch_single_device.json 2> gpurun_out/r3_tp2_self_launch.err
echo "self-launch rc=$?"; python - <<PY
import json
try:
    d=json.loads(open("gpurun_out/r3_tp2_self_launch_single_device.json").read().strip().splitlines()[-1])
    print("tp2", d["n_gpus"], d["value"], d["parity_checked"], d["config"].get("tp_exchange"), d["config"].get("decode_graph"))
except Exception as e:
    print("FAILED", e); print(open("gpurun_out/r3_tp2_self_launch.err").read()[-1500:])
PY
timeout 600 python tools/llm_tp_spawn_check.py > gpurun_out/r3_llm_tp2_spawn_single_device.json 2> gpurun_out/r3_llm_tp2_spawn.err
echo "llm spawn rc=$?"; tail -c 600 gpurun_out/r3_llm_tp2_spawn_single_device.json
timeout 400 python -m pytest tests/test_gpu_parity_real_size.py -q -x -m gpu -k "rope_in" 2>&1 | tail -2
