#!/bin/bash
R="${GRAFT_REPO_ROOT:-.}"; cd /tmp; export TMPDIR=/tmp; rm -rf /tmp/kt5
timeout 400 rocprofv3 --kernel-trace -d /tmp/kt5 -o p -- python $R/bench.py --batch 16 --weights fp8 --series 8 --length 1024 --steps 8 --warmup 2 --no-cpu-baseline > /tmp/kt5.log 2>&1
db=$(find /tmp/kt5 -name "*.db" | head -1)
mkdir -p $R/gpurun_out
(echo "## rocprofv3 --kernel-trace -- python bench.py --batch 16 --weights fp8 --series 8 --length 1024 --steps 8 --warmup 2 --no-cpu-baseline  (MI355X, round 3, 16 attention slots + tile prefetch)"; python $R/tools/prof_db.py $db) > $R/gpurun_out/r3_cfg5_kernel_trace.txt 2>&1
grep -v "fill_hash\|gemm_dma\|attn_prefill\|rope_kv\|embed_merge\|ts_pat" $R/gpurun_out/r3_cfg5_kernel_trace.txt | head -30 | cut -c1-200
