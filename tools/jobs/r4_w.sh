#!/bin/bash
# planes attention with swizzled LDS rows: tests, kernel time, LDS counters, TTFT
cd /root/repo; mkdir -p gpurun_out
python -m pytest tests/test_gpu_kernels.py -x -q -m gpu -k "planes" 2>&1 | tail -2
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/ktw; rocprofv3 --kernel-trace -d /tmp/ktw -o p -- python /root/repo/tools/attn_prefill_time.py > /dev/null 2>&1
python /root/repo/tools/prof_db.py $(find /tmp/ktw -name "*.db" | head -1) 2>/dev/null | grep -i "attn_prefill\|kv_planes" | cut -c1-150 | tee /root/repo/gpurun_out/r4_attn_planes_swizzle.txt
rm -rf /tmp/pm; rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_INST_LDS SQ_INSTS_LDS -d /tmp/pm -o p -- python /root/repo/tools/attn_prefill_time.py > /dev/null 2>&1
python /root/repo/tools/prof_db.py $(find /tmp/pm -name "*.db" | head -1) 2>/dev/null | grep -i "attn_prefill_planes" | cut -c1-150 | tee -a /root/repo/gpurun_out/r4_attn_planes_swizzle.txt
cd /root/repo
for i in 1 2; do python bench.py --steps 4 --warmup 2 --no-cpu-baseline --ttft-runs 7 2>/dev/null | tail -1 | python -c "
import json,sys; r=json.loads(sys.stdin.read()); print('ttft', r['ttft_ms_p50'], r['parity_checked'])"; done | tee -a gpurun_out/r4_attn_planes_swizzle.txt
