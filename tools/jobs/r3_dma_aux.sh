#!/bin/bash
R="${GRAFT_REPO_ROOT:-.}"; cd $R
for lib in "" chatts_amd/lib/alt_dma_A.so chatts_amd/lib/alt_dma_W.so ""; do
export CHATTS_AMD_LIB=$lib
timeout 300 python bench.py --steps 8 --warmup 2 --no-cpu-baseline --ttft-runs 9 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('lib ${lib:-default}: ttft', round(d['ttft_ms_p50'],2), 'prefill gate_up us', round(d['prefill_roofline']['avg_us'],1), d['parity_checked'])"
done
