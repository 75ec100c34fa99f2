#!/bin/bash
# round 6, job p: per-kernel durations of the bf16 headline decode step under the GEMV geometries (rows x chunks in flight)
R="${GRAFT_REPO_ROOT:-.}"; O=$R/gpurun_out/r6_p; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
i=0
for cfg in "X=0" "CHATTS_GEMV_ROWS=4 CHATTS_GEMV_UNR=2" "CHATTS_GEMV_ROWS=2 CHATTS_GEMV_UNR=4" "CHATTS_GEMV_ROWS=4 CHATTS_GEMV_UNR=4"; do
  i=$((i+1))
  rm -rf /tmp/ktp$i
  env $cfg timeout 400 rocprofv3 --kernel-trace -d /tmp/ktp$i -o p -- python $R/bench.py --steps 40 --warmup 5 --no-cpu-baseline --ttft-runs 1 > /tmp/ktp$i.log 2>&1
  db=$(find /tmp/ktp$i -name "*.db" | head -1)
  echo "== $cfg   $(tail -1 /tmp/ktp$i.log | python -c 'import json,sys; d=json.loads(sys.stdin.read()); print("tok/s under the profiler", round(d["value"],1))' 2>/dev/null)" >> $O/traces.txt
  python $R/tools/prof_db.py $db | grep "gemv_ldsx\|calls" | head -8 | cut -c1-150 >> $O/traces.txt
done
cat $O/traces.txt
