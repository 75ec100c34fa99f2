#!/bin/bash
R="${GRAFT_REPO_ROOT:-.}"; cd $R
for mb in 64 128 64 128; do
CHATTS_GEMM_STREAM_MB=$mb timeout 200 python bench.py --model chatts-8b --series 1 --length 256 --steps 16 --warmup 4 --no-cpu-baseline --ttft-runs 15 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('8B cfg2 stream_mb $mb: ttft', round(d['ttft_ms_p50'],3), 'tok/s', round(d['value'],1), d['parity_checked'])"
done
