#!/bin/bash
# round 6, job u: bf16 decode GEMV waves per workgroup / workgroups per CU, per-kernel durations in the headline bench (kernel trace per setting)
R="${GRAFT_REPO_ROOT:-.}"; O=$R/gpurun_out/r6_u; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
i=0
for cfg in "X=0" "CHATTS_GEMV_NW=4" "CHATTS_GEMV_NW=8" "CHATTS_GEMV_NW=12" "CHATTS_GEMV_NW=16" "CHATTS_GEMV_OCC=2" "CHATTS_GEMV_OCC=3"; do
  i=$((i+1)); rm -rf /tmp/ktu$i
  env $cfg timeout 400 rocprofv3 --kernel-trace -d /tmp/ktu$i -o p -- python $R/bench.py --steps 40 --warmup 5 --no-cpu-baseline --ttft-runs 1 > /tmp/ktu$i.log 2>&1
  echo "== $cfg" >> $O/traces.txt
  python $R/tools/prof_db.py $(find /tmp/ktu$i -name "*.db" | head -1) | grep "gemv_ldsx" | head -5 | cut -c1-140 >> $O/traces.txt
done
cat $O/traces.txt
