#!/bin/bash
# the whole GPU suite and the driver smoke test on a gpurun box:  gpurun --timeout 1800 -- bash tools/jobs/full_validation.sh
mkdir -p gpurun_out
timeout 1200 python -m pytest tests -m gpu -x -q --timeout 300 2>&1 | tail -4
timeout 120 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
