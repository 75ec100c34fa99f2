#!/bin/bash
# config 5 FETCH_SIZE pass again (the first tool version picked lm_head instead of gate_up) + bulk all-reduce grid sweep at W = 8
R="${GRAFT_REPO_ROOT:-.}"; O=$R/gpurun_out; mkdir -p $O
export PMC_TRAFFIC_OUT=$O/pmc_traffic.json HSA_ENABLE_IPC_MODE_LEGACY=0
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/fs5; C5="python bench.py --batch 16 --weights fp8 --series 8 --length 1024 --steps 4 --warmup 2 --no-cpu-baseline"
timeout 500 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d /tmp/fs5 -o p -- python $R/bench.py --batch 16 --weights fp8 --series 8 --length 1024 --steps 4 --warmup 2 --no-cpu-baseline > /tmp/fs5.log 2>&1
db=$(find /tmp/fs5 -name "*.db" | head -1)
(echo "## rocprofv3 --kernel-trace --pmc FETCH_SIZE -- $C5   (MI355X, round 4, final code)"; python $R/tools/prof_db.py $db | grep -v fill_hash | grep "gemm_stream\|attn_decode\|splitk\|rmsnorm\|kernel-trace\|PMC\|calls") > $O/r4_cfg5_pmc_fetch_size.txt
python $R/tools/pmc_traffic.py batched $db "rocprofv3 --kernel-trace --pmc FETCH_SIZE -- $C5" profiles/r4_cfg5_pmc_fetch_size.txt | cut -c1-900
cd $R
for b in 32 64 128 256; do
  CHATTS_TP_BULK_BLOCKS=$b timeout 300 python tools/tp_shard_step.py --worlds 8 --steps 8 --warmup 2 --prefill-runs 3 --out $O/r4n_bulk$b.json > /dev/null 2> $O/r4n_bulk$b.err
  echo "bulk blocks $b: $(grep tp_shard_step $O/r4n_bulk$b.err | sed 's/.*prefill_ms": \([0-9.]*\).*/prefill_ms \1/')"
done
timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('bench traffic', d['roofline']['traffic'], d['roofline']['traffic_source'][:80], '| ts', d['ts_encoder_roofline'].get('traffic'))"
