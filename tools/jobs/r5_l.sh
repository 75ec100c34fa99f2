#!/bin/bash
# round 5, after the final evidence job: the in-process two-rank tests twice (which one skipped?), the conservative-fence W = 8 rank
# with its own cap, and the headline / config 5 bench lines with the committed FETCH_SIZE passes in place (traffic non-null)
R="${GRAFT_REPO_ROOT:-.}"; O=$R/gpurun_out/r5_l; mkdir -p $O; cd $R
export HSA_ENABLE_IPC_MODE_LEGACY=0
for i in 1 2; do timeout 600 python -m pytest tests/test_gpu_tp_p2p.py -q -m gpu -rs > $O/pytest_p2p_$i.txt 2>&1; tail -4 $O/pytest_p2p_$i.txt; done
CHATTS_TP_BULK_FENCE=1 timeout 300 python tools/tp_shard_step.py --worlds 8 --out $O/r5_tp_shard_step_w8_threadfence.json > /dev/null 2> $O/fence.err
python -c "
import json; v=json.load(open('$O/r5_tp_shard_step_w8_threadfence.json'))['worlds']['8']; print('W=8 conservative fence: prefill %.2f ms decode %.3f ms' % (v['prefill_ms'], v['decode_ms_per_step']))"
timeout 600 python bench.py --steps 32 --warmup 8 > $O/r5_bench_n1.json 2> $O/n1.err
timeout 400 python bench.py --batch 16 --weights fp8 --series 8 --length 1024 --steps 24 --warmup 6 --no-cpu-baseline > $O/r5_bench_cfg5_fp8_8x1024_b16.json 2> $O/cfg5.err
for f in r5_bench_n1 r5_bench_cfg5_fp8_8x1024_b16; do python - $O/$f.json <<'PY'
import json, sys
r = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(sys.argv[1].split("/")[-1], "value %.1f" % r["value"], "ms/step %.3f" % r["ms_per_step"], "ttft", r.get("ttft_ms_p50"), "parity_checked", r.get("parity_checked"),
      "roofline", {k: r["roofline"].get(k) for k in ("achieved", "frac", "traffic", "avg_us")}, "ts", (r.get("ts_encoder_roofline") or {}).get("traffic"))
PY
done
