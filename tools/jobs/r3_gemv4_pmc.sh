#!/bin/bash
R="${GRAFT_REPO_ROOT:-.}"; cd /tmp; export TMPDIR=/tmp; mkdir -p $R/gpurun_out
timeout 200 python $R/tools/gemv4_prof.py 5 2>&1 | grep -v amdgpu.ids | tail -4
rm -rf /tmp/g4
timeout 300 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVES SQ_INSTS_VMEM_RD SQ_INSTS_LDS -d /tmp/g4 -o p -- python $R/tools/gemv4_prof.py 3 > /tmp/g4.log 2>&1
db=$(find /tmp/g4 -name "*.db" | head -1)
(echo "## rocprofv3 --kernel-trace --pmc SQ_* -- python tools/gemv4_prof.py 3   (MI355X, round 3: the int4 gate_up GEMV next to the bf16 one)"; python $R/tools/prof_db.py $db) > $R/gpurun_out/r3_pmc_gemv4.txt 2>&1
grep "gemv" $R/gpurun_out/r3_pmc_gemv4.txt | cut -c1-150 | head -24
