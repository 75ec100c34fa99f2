#!/bin/bash
R="${GRAFT_REPO_ROOT:-.}"; cd "$R"; mkdir -p gpurun_out
timeout 300 python -m pytest tests/test_gpu_decode_mega.py -x -q --timeout 120 2>&1 | tail -15 > gpurun_out/r3_mega2_pytest.log
tail -6 gpurun_out/r3_mega2_pytest.log
timeout 300 python tools/mega_check.py --steps 32 > gpurun_out/r3_mega2_full.json 2> gpurun_out/r3_mega2_full.err; echo "full rc=$?"; tail -c 900 gpurun_out/r3_mega2_full.json; tail -2 gpurun_out/r3_mega2_full.err
timeout 300 python tools/mega_profile.py > gpurun_out/r3_mega_profile2.json 2> gpurun_out/r3_mega_profile2.err; python - <<'PY'
import json
d=json.load(open("gpurun_out/r3_mega_profile2.json"))
print("step_us", d["step_us"], "per_layer", d["per_layer_us_wg0"])
for k,v in d["phases"].items():
    if k.startswith("wg0"): print(k, v)
PY
