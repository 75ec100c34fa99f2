#!/bin/bash
timeout 300 python -m chatts_amd.build > /dev/null 2>&1
bash tools/jobs/tp2_single_device.sh 2>&1 | tail -12
cp gpurun_out/r2_tp2_single_device.json gpurun_out/r2_tp2_normal.json
export CHATTS_BENCH_INJECT_P2P_STALL=1
bash tools/jobs/tp2_single_device.sh 2>&1 | tail -12
cp gpurun_out/r2_tp2_single_device.json gpurun_out/r2_tp2_injected_stall.json
cp gpurun_out/r2_tp2_normal.json gpurun_out/r2_tp2_single_device.json
