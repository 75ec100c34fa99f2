#!/bin/bash
R="${GRAFT_REPO_ROOT:-.}"; cd "$R"; mkdir -p gpurun_out
timeout 300 python -m pytest tests/test_gpu_decode_mega.py -x -q --timeout 120 2>&1 | tail -15 > gpurun_out/r3_mega3_pytest.log
tail -4 gpurun_out/r3_mega3_pytest.log
for D in 10 16; do
CHATTS_MEGA_DEPTH=$D timeout 300 python tools/mega_profile.py > gpurun_out/r3_mega_profile_d$D.json 2> gpurun_out/r3_mega_profile_d$D.err; D=$D python - <<'PY'
import json, os
d=json.load(open("gpurun_out/r3_mega_profile_d%s.json" % os.environ["D"]))
print("depth", os.environ["D"], "step_us", round(d["step_us"],1), "per_layer", round(d["per_layer_us_wg0"],2))
for k,v in d["phases"].items():
    if k.startswith("wg0"): print(" ", k, {kk:vv for kk,vv in v.items() if kk not in ("to_B","wave0_start_after_phase_start")})
PY
done
