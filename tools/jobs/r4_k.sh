#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out; O=gpurun_out; R=$PWD
timeout 600 python -m pytest tests/test_gpu_kernels.py -m gpu -q -k "fp8_speed or quantize_rows_fp8" 2>&1 | tail -3
python tools/fp8_gemm_prof.py
cd /tmp && export TMPDIR=/tmp; rm -rf /tmp/pf
REPS=3 timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE -d /tmp/pf -o p -- python $R/tools/fp8_gemm_prof.py > /tmp/pf.log 2>&1
(echo "## rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE -- python tools/fp8_gemm_prof.py  (REPS=3, MI355X, round 4)"; python $R/tools/prof_db.py $(find /tmp/pf -name "*.db" | head -1) | grep "gemm_fp8\|counter\|PMC") > $R/$O/r4_pmc_gemm_fp8.txt
grep "gemm_fp8_kernel<1" $R/$O/r4_pmc_gemm_fp8.txt | cut -c1-140
