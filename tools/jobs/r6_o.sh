#!/bin/bash
# round 6, job o: kernel trace of config 5 (16 sequences, fp8 weights, 8 x 1024) on the current code
R="${GRAFT_REPO_ROOT:-.}"; O=$R/gpurun_out/r6_o; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/kt5
timeout 600 rocprofv3 --kernel-trace -d /tmp/kt5 -o p -- python $R/bench.py --batch 16 --weights fp8 --series 8 --length 1024 --steps 16 --warmup 4 --no-cpu-baseline > /tmp/kt5.log 2>&1
tail -1 /tmp/kt5.log | cut -c1-300
python $R/tools/prof_db.py $(find /tmp/kt5 -name "*.db" | head -1) | grep -v fill_hash | head -40 | cut -c1-200 > $O/cfg5_trace.txt
cat $O/cfg5_trace.txt
