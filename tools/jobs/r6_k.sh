#!/bin/bash
# round 6, job k: per-kernel durations of the fp8-weight decode step under the GEMV8 geometries (kernel trace of the bench)
R="${GRAFT_REPO_ROOT:-.}"; O=$R/gpurun_out/r6_k; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
i=0
for cfg in "X=0" "CHATTS_GEMV8_ROWS=4 CHATTS_GEMV8_UNR=2" "CHATTS_GEMV8_ROWS=4 CHATTS_GEMV8_UNR=1" "CHATTS_GEMV8_ROWS=2 CHATTS_GEMV8_UNR=4"; do
  i=$((i+1))
  rm -rf /tmp/kt$i
  env $cfg timeout 400 rocprofv3 --kernel-trace -d /tmp/kt$i -o p -- python $R/bench.py --weights fp8 --steps 40 --warmup 5 --no-cpu-baseline --ttft-runs 1 > /tmp/kt$i.log 2>&1
  db=$(find /tmp/kt$i -name "*.db" | head -1)
  echo "== $cfg" >> $O/traces.txt
  python $R/tools/prof_db.py $db | grep "gemv8\|attn_decode\|calls" | head -14 | cut -c1-150 >> $O/traces.txt
done
cat $O/traces.txt
