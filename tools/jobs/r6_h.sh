#!/bin/bash
# round 6, job h: (T, split-K) of the residual projections at M = 798 - the cost model's pick against its runners-up, whole call (GEMM + split-K epilogue + post-norm planes)
R="${GRAFT_REPO_ROOT:-.}"; O=$R/gpurun_out/r6_h; mkdir -p $O
cd $R
for cfg in "0 0" "5 5" "6 2" "6 4" "5 2" "6 3" "7 2" "5 4"; do
  set -- $cfg
  echo "== down GEMM_T=$1 GEMM_SK=$2" >> $O/sweep.txt
  CHATTS_GEMM_T=$1 CHATTS_GEMM_SK=$2 TILED_SHAPES=down timeout 300 python tools/tiled_check.py 798 5 2>&1 | grep "^down .*median" | cut -c1-220 >> $O/sweep.txt
done
for cfg in "0 0" "6 2" "12 1" "10 1" "6 1" "5 2" "6 3"; do
  set -- $cfg
  echo "== o GEMM_T=$1 GEMM_SK=$2" >> $O/sweep.txt
  CHATTS_GEMM_T=$1 CHATTS_GEMM_SK=$2 TILED_SHAPES=o timeout 300 python tools/tiled_check.py 798 5 2>&1 | grep "^o .*median" | cut -c1-220 >> $O/sweep.txt
done
for cfg in "0 0" "9 1" "8 1" "7 1" "6 1" "5 2"; do
  set -- $cfg
  echo "== qkv GEMM_T=$1 GEMM_SK=$2" >> $O/sweep.txt
  CHATTS_GEMM_T=$1 CHATTS_GEMM_SK=$2 TILED_SHAPES=qkv timeout 300 python tools/tiled_check.py 798 5 2>&1 | grep "^qkv .*median" | cut -c1-220 >> $O/sweep.txt
done
for cfg in "0 0" "7 1" "6 1" "5 1" "8 1"; do
  set -- $cfg
  echo "== gate_up GEMM_T=$1 GEMM_SK=$2" >> $O/sweep.txt
  CHATTS_GEMM_T=$1 CHATTS_GEMM_SK=$2 TILED_SHAPES=gate_up timeout 300 python tools/tiled_check.py 798 5 2>&1 | grep "^gate_up .*median" | cut -c1-220 >> $O/sweep.txt
done
cat $O/sweep.txt
