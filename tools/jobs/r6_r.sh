#!/bin/bash
# round 6, job r: config 5 (16 x fp8) per-kernel durations under the weight-stream geometries (waves per workgroup, ring depth)
R="${GRAFT_REPO_ROOT:-.}"; O=$R/gpurun_out/r6_r; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
i=0
for cfg in "X=0" "CHATTS_GEMM_STREAM_WAVES=8" "CHATTS_GEMM_STREAM_STAGES=3" "CHATTS_GEMM_STREAM_WAVES=8 CHATTS_GEMM_STREAM_STAGES=3"; do
  i=$((i+1)); rm -rf /tmp/ktr$i
  env $cfg timeout 500 rocprofv3 --kernel-trace -d /tmp/ktr$i -o p -- python $R/bench.py --batch 16 --weights fp8 --series 8 --length 1024 --steps 16 --warmup 4 --no-cpu-baseline > /tmp/ktr$i.log 2>&1
  echo "== $cfg   $(tail -1 /tmp/ktr$i.log | python -c 'import json,sys; d=json.loads(sys.stdin.read()); print("tok/s under the profiler", round(d["value"],1), "ms/step", round(d["ms_per_step"],3))' 2>/dev/null)" >> $O/traces.txt
  python $R/tools/prof_db.py $(find /tmp/ktr$i -name "*.db" | head -1) | grep "gemm_stream\|attn_decode\|epilogue_norm_q\|epilogue_v4" | head -9 | cut -c1-150 >> $O/traces.txt
done
cat $O/traces.txt
