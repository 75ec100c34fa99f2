#!/bin/bash
# round 6: the whole GPU suite + smoke on the current code
R="${GRAFT_REPO_ROOT:-.}"; O=$R/gpurun_out/r6_full; mkdir -p $O
cd $R
timeout 2400 python -m pytest tests/ -x -q -m gpu > $O/pytest_gpu.txt 2>&1
tail -6 $O/pytest_gpu.txt | cut -c1-300
timeout 600 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.txt 2>&1
tail -2 $O/smoke.txt | cut -c1-300
