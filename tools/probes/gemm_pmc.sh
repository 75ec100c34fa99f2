# PMC passes over one prefill GEMM shape (LDS-DMA kernel).  usage: bash tools/probes/gemm_pmc.sh <shape> <out-name>
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
SHAPE=${1:-gate_up}; OUT=$R/gpurun_out/${2:-gemm_dma_pmc}.txt
rm -f $OUT
i=0
for pmc in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_BUSY_CU_CYCLES" "SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_INSTS_VMEM SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_PENDING_STALL_CYCLES_sum GRBM_GUI_ACTIVE" "TCC_REQ_sum TCC_EA0_RDREQ_sum TCC_HIT_sum TCC_MISS_sum"; do
  i=$((i+1))
  timeout 120 rocprofv3 --kernel-trace --pmc $pmc -d /tmp/pp$i -o p -- python $R/tools/gemm_prof.py $SHAPE 798 planes > /tmp/pp$i.log 2>&1
  db=$(find /tmp/pp$i -name "*.db" | head -1)
  echo "## pmc: $pmc" >> $OUT
  if [ -n "$db" ]; then python $R/tools/prof_db.py $db | grep -v "fill_hash\|^#\|^$" >> $OUT; else tail -3 /tmp/pp$i.log >> $OUT; fi
done
