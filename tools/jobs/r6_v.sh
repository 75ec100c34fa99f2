#!/bin/bash
# round 6, job v: SQ / TCP / TCC counter passes over the prefill kernel at gate_up, M = 798: row-major against tiled weights (what do the
# waves wait for; how many L2 requests per byte)
R="${GRAFT_REPO_ROOT:-.}"; O=$R/gpurun_out/r6_v; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for mode in planes tiled; do
  i=0
  for pmc in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES" "SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_VMEM SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_PENDING_STALL_CYCLES_sum GRBM_GUI_ACTIVE" "TCC_REQ_sum TCC_EA0_RDREQ_sum TCC_HIT_sum TCC_MISS_sum"; do
    i=$((i+1)); rm -rf /tmp/pv$mode$i
    timeout 120 rocprofv3 --kernel-trace --pmc $pmc -d /tmp/pv$mode$i -o p -- python $R/tools/gemm_prof.py gate_up 798 $mode > /tmp/pv$mode$i.log 2>&1
    db=$(find /tmp/pv$mode$i -name "*.db" | head -1)
    echo "## W $mode ($([ $mode = planes ] && echo row-major || echo tiled)); pmc: $pmc" >> $O/ring_pmc.txt
    if [ -n "$db" ]; then python $R/tools/prof_db.py $db | grep "gemm_ring" | grep -v "^#" | cut -c1-150 >> $O/ring_pmc.txt; else tail -3 /tmp/pv$mode$i.log >> $O/ring_pmc.txt; fi
  done
done
cat $O/ring_pmc.txt
