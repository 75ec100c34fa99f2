#!/bin/bash
R="${GRAFT_REPO_ROOT:-.}"; cd $R; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_decode_mega.py -q -x -m gpu -k "attn or attention or decode or mega or persistent" 2>&1 | tail -4
timeout 900 python -m pytest tests/test_gpu_e2e.py tests/test_gpu_parity_real_size.py -q -x -m gpu 2>&1 | tail -3
timeout 300 python bench.py --batch 16 --weights fp8 --series 8 --length 1024 --steps 24 --warmup 6 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('cfg5', d['value'], d['ms_per_step'], d['parity_checked'])"
timeout 300 python bench.py --steps 32 --warmup 8 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('n1', d['value'], d['ms_per_step'], d['ttft_ms_p50'], d['parity_checked'])"
timeout 300 python bench.py --series 30 --lengths mixed --steps 32 --warmup 8 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('cfg4', d['value'], d['ms_per_step'], d['ttft_ms_p50'], d['parity_checked'])"
