#!/bin/bash
R="${GRAFT_REPO_ROOT:-.}"
cd /tmp && export TMPDIR=/tmp
mkdir -p $R/gpurun_out
for C in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVES" "SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_VMEM GRBM_GUI_ACTIVE"; do
  rm -rf /tmp/pm
  timeout 200 rocprofv3 --kernel-trace --pmc $C -d /tmp/pm -o p -- python $R/tools/gemv4_prof.py 2 > /tmp/pm.log 2>&1
  db=$(find /tmp/pm -name "*.db" | head -1)
  python $R/tools/prof_db.py $db | grep -i "gemv\|calls\|PMC" >> $R/gpurun_out/r2_pmc_gemv4.txt
  tail -3 /tmp/pm.log | cut -c1-200
done
cat $R/gpurun_out/r2_pmc_gemv4.txt | cut -c1-220
