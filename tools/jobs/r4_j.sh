#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out; O=gpurun_out; R=$PWD
python tools/fp8_gemm_prof.py
cd /tmp && export TMPDIR=/tmp
for pm in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM SQ_BUSY_CYCLES"; do
  rm -rf /tmp/pf
  REPS=3 timeout 300 rocprofv3 --kernel-trace --pmc $pm -d /tmp/pf -o p -- python $R/tools/fp8_gemm_prof.py > /tmp/pf.log 2>&1
  python $R/tools/prof_db.py $(find /tmp/pf -name "*.db" | head -1) | grep -A40 "PMC pass" | grep "gemm_fp8\|counter" | cut -c1-150
done
