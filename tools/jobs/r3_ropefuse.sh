#!/bin/bash
R="${GRAFT_REPO_ROOT:-.}"; cd $R; mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_parity_real_size.py -q -x -m gpu -k "rope_in or bench_prompt" 2>&1 | tail -4
timeout 600 python -m pytest tests/test_gpu_e2e.py tests/test_gpu_kernels.py -q -x -m gpu -k "prefill or rope or chunk or paged or packed or post_norm" 2>&1 | tail -3
for f in 0 1; do
CHATTS_ROPE_FUSE=$f timeout 300 python bench.py --steps 16 --warmup 4 --no-cpu-baseline --ttft-runs 7 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('rope fuse $f: ttft', d['ttft_ms_p50'], 'tok/s', d['value'], 'parity', d['parity_checked'])"
done
