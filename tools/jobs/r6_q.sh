#!/bin/bash
# round 6, job q: kernel trace of the TS encoder alone (8 x 256), fused layer 0 vs the two-launch form
R="${GRAFT_REPO_ROOT:-.}"; O=$R/gpurun_out/r6_q; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for f in 1 0; do
  rm -rf /tmp/ktt$f
  ( cd $R && CHATTS_TS_L0_FUSED=$f TS_CALLS=20 timeout 300 rocprofv3 --kernel-trace -d /tmp/ktt$f -o p -- python tools/pmc_traffic.py ts-run > /tmp/ktt$f.log 2>&1 )
  echo "== TS_L0_FUSED=$f" >> $O/ts_trace.txt
  python $R/tools/prof_db.py $(find /tmp/ktt$f -name "*.db" | head -1) | grep -v fill_hash | head -12 | cut -c1-170 >> $O/ts_trace.txt
done
cat $O/ts_trace.txt
