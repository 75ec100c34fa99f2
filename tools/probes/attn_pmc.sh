# PMC passes over the prefill attention kernel.  usage: bash tools/probes/attn_pmc.sh <out-name>
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/${1:-attn_pmc}.txt
rm -f $OUT
i=0
for pmc in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_BUSY_CU_CYCLES" "SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "SQ_INSTS_VMEM SQ_ACTIVE_INST_VMEM SQ_INSTS_SALU SQ_INSTS_MFMA SQ_WAVES GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  timeout 120 rocprofv3 --kernel-trace --pmc $pmc -d /tmp/ap$i -o p -- python $R/tools/attn_prefill_time.py > /tmp/ap$i.log 2>&1
  db=$(find /tmp/ap$i -name "*.db" | head -1)
  echo "## pmc: $pmc" >> $OUT
  if [ -n "$db" ]; then python $R/tools/prof_db.py $db | grep -v "fill_hash\|^#\|^$" >> $OUT; else tail -3 /tmp/ap$i.log >> $OUT; fi
done
