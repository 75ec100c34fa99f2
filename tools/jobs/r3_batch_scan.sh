#!/bin/bash
R="${GRAFT_REPO_ROOT:-.}"; cd $R; mkdir -p gpurun_out
echo "## bench.py --batch B --weights W --series 8 --length 256 --steps 24 --warmup 6 --no-cpu-baseline (MI355X, round 3 final code): aggregate tokens/s, ms per B-wide step" > gpurun_out/r3_batched_decode.txt
for w in bf16 fp8; do for b in 2 4 8 16; do
timeout 200 python bench.py --batch $b --weights $w --series 8 --length 256 --steps 24 --warmup 6 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('weights $w  B = %2d  %8.1f tok/s  %6.3f ms/step  ttft p50 %6.1f ms' % ($b, d['value'], d['ms_per_step'], d['ttft_ms_p50']))" | tee -a gpurun_out/r3_batched_decode.txt
done; done
