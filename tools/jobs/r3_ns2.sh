#!/bin/bash
R="${GRAFT_REPO_ROOT:-.}"; cd $R
for ns in 8 12 16 24 32; do
CHATTS_BATCH_NSPLITS=$ns timeout 300 python bench.py --batch 16 --weights fp8 --series 8 --length 1024 --steps 24 --warmup 6 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('nsplits $ns', d['value'], d['ms_per_step'], d['parity_checked'])"
done
