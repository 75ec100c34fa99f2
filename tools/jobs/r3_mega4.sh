#!/bin/bash
R="${GRAFT_REPO_ROOT:-.}"; cd "$R"; mkdir -p gpurun_out
CHATTS_MEGA_DEPTH=10 timeout 300 python tools/mega_profile.py > gpurun_out/r3_mega_profile_spread.json 2> gpurun_out/r3_mega_profile_spread.err
python - <<'PY'
import json
d=json.load(open("gpurun_out/r3_mega_profile_spread.json"))
print("step_us", d["step_us"], "per_layer", d["per_layer_us_wg0"])
for k,v in d["finish_spread_layer1"].items(): print(k, v)
PY
tail -3 gpurun_out/r3_mega_profile_spread.err
