#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out
timeout 420 python tools/tp2_server_check.py > gpurun_out/r2_tp2_server.out 2>&1
tail -c 1500 gpurun_out/r2_tp2_server.out; tail -5 gpurun_out/r2_tp2_server.log
