#!/bin/bash
# what would a narrower KV cache cost in logits error?  K / V values rounded on the way into the float32 cache (CHATTS_KV_ROUND), the
# HIP path against the float32 full-depth CPU oracle on the headline workload (8 x 256, ChatTS-14B, 48 layers)
R="${GRAFT_REPO_ROOT:-.}"; cd $R; mkdir -p gpurun_out
for m in 1 3 2; do
CHATTS_KV_ROUND=$m timeout 900 python tools/parity_full_depth.py --out gpurun_out/r3_kv_round_$m.json > gpurun_out/r3_kv_round_$m.log 2>&1
python - <<PY
import json
d=json.load(open("gpurun_out/r3_kv_round_$m.json"))
print("kv_round $m", {k: d.get(k) for k in ("max_step_logits_rel_err","max_abs_err_over_max_logit","tokens_match","ok")}, "gpu", d.get("tokens_gpu"), "oracle", d.get("tokens_oracle"))
PY
done
