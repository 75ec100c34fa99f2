#!/bin/bash
R="${GRAFT_REPO_ROOT:-.}"; cd $R
for lib in "" chatts_amd/lib/alt_kvnt.so; do
export CHATTS_AMD_LIB=$lib
echo "== lib: ${lib:-default}"
timeout 300 python bench.py --batch 16 --weights fp8 --series 8 --length 1024 --steps 24 --warmup 6 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('cfg5', d['value'], d['ms_per_step'], d['parity_checked'])"
timeout 300 python bench.py --series 30 --lengths mixed --steps 32 --warmup 8 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('cfg4', d['value'], d['ms_per_step'], d['parity_checked'])"
timeout 300 python bench.py --steps 32 --warmup 8 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('n1', d['value'], d['ms_per_step'], d['parity_checked'])"
done
