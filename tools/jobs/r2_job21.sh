#!/bin/bash
R="${GRAFT_REPO_ROOT:-.}"
cd /tmp && export TMPDIR=/tmp
mkdir -p $R/gpurun_out
rm -f $R/gpurun_out/r2_pmc_attn_bf16x3.txt
for C in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVES" "SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_WAIT_INST_LDS"; do
  rm -rf /tmp/pm
  timeout 200 rocprofv3 --kernel-trace --pmc $C -d /tmp/pm -o p -- python $R/tools/attn_prefill_time.py > /tmp/pm.log 2>&1
  db=$(find /tmp/pm -name "*.db" | head -1)
  python $R/tools/prof_db.py $db | grep -i "attn_prefill\|calls\|PMC" >> $R/gpurun_out/r2_pmc_attn_bf16x3.txt
done
cat $R/gpurun_out/r2_pmc_attn_bf16x3.txt | cut -c1-200
