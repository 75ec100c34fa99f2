#!/bin/bash
# round 6, job a: the two gates of the 1.5-pass prefill split + a baseline line of this box
#  (ii) matrix-pipe rate / clock of the candidate instruction mixes on GEMM-like operand data (tools/probes/mfma_mix_probe.hip)
#  (i)  full-depth numerics of the candidate splits, emulated in torch float64 (tools/split_emulation.py)
R="${GRAFT_REPO_ROOT:-.}"; O=$R/gpurun_out/r6_a; mkdir -p $O
cd $R
(echo "## round 6: tools/probes/mfma_mix_probe (MI355X): per (16x16 tile, 128 K-values) instruction mixes on GEMM-like operand data, 2 waves per SIMD, registers only"
 ./tools/probes/mfma_mix_probe) > $O/r6_mfma_mix_probe.txt 2>&1
cat $O/r6_mfma_mix_probe.txt
timeout 900 python tools/split_emulation.py --out $O/r6_split_emulation.json > $O/split_emulation.log 2>&1
tail -5 $O/split_emulation.log | cut -c1-400
timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_baseline.json 2> $O/bench_baseline.err
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r6_a/bench_baseline.json").read().strip().splitlines()[-1])
print("baseline: tok/s", d["value"], "ttft", d["ttft_ms_p50"], "ts_us", d["ts_encoder_roofline"]["avg_us"], "gate_up prefill us", d["prefill_roofline"]["avg_us"])
PY
