#!/bin/bash
R="${GRAFT_REPO_ROOT:-.}"; cd /tmp; export TMPDIR=/tmp; rm -rf /tmp/kt
timeout 300 rocprofv3 --kernel-trace -d /tmp/kt -o p -- python $R/bench.py --steps 8 --warmup 2 --no-cpu-baseline --ttft-runs 1 > /tmp/kt.log 2>&1
db=$(find /tmp/kt -name "*.db" | head -1)
python $R/tools/prof_db.py $db | grep "attn_decode\|gemv_ldsx\|argmax" | cut -c1-150
tail -1 /tmp/kt.log | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('n1', d['value'], d['ms_per_step'])"
