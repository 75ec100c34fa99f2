#!/bin/bash
R="${GRAFT_REPO_ROOT:-.}"; cd $R; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_kernels.py -q -x -m gpu -k "gemm or linear or stream or ts_encode or split" 2>&1 | tail -3
timeout 200 python tools/ts_gemm_sweep.py 128 2>&1 | grep -v amdgpu.ids | grep "auto\|8 waves$\|== P" | head -8
timeout 300 python bench.py --steps 16 --warmup 4 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('n1', d['value'], d['ttft_ms_p50'], d['parity_checked'], 'ts', d['ts_encoder_roofline']['avg_us'], d['ts_encoder_roofline']['frac'])"
timeout 300 python bench.py --batch 16 --weights fp8 --series 8 --length 1024 --steps 24 --warmup 6 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('cfg5', d['value'], d['ms_per_step'], d['parity_checked'])"
