#!/bin/bash
R="${GRAFT_REPO_ROOT:-.}"; cd $R
timeout 300 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_decode_mega.py -q -x -m gpu -k "attn_decode or attention_decode or mega or persistent or decode_attention" 2>&1 | tail -2
timeout 300 python bench.py --batch 16 --weights fp8 --series 8 --length 1024 --steps 24 --warmup 6 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('cfg5', d['value'], d['ms_per_step'], d['parity_checked'])"
timeout 300 python bench.py --steps 32 --warmup 8 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('n1', d['value'], d['ms_per_step'], d['parity_checked'])"
