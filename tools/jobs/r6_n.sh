#!/bin/bash
# round 6, job n: one TS-encoder MLP layer (M = 128, 5120 x 5120, GELU -> planes): the shipped multi-block stream kernel against the prefill
# kernel on tiled operands at several split-K factors (whole chatts_linear call incl. the split-K epilogue)
R="${GRAFT_REPO_ROOT:-.}"; O=$R/gpurun_out/r6_n; mkdir -p $O
cd $R
echo "== stream (shipped)" >> $O/ts_layer.txt
TILED_SHAPES=ts TILED_VARIANTS=rm timeout 300 python tools/tiled_check.py 128 7 2>&1 | grep "^ts .*median" | cut -c1-220 >> $O/ts_layer.txt
for sk in 0 6 8 10 12 16; do
  echo "== ring, GEMM_SK=$sk" >> $O/ts_layer.txt
  CHATTS_GEMM_STREAM_MB=16 CHATTS_GEMM_SK=$sk TILED_SHAPES=ts timeout 300 python tools/tiled_check.py 128 7 2>&1 | grep "^ts " | cut -c1-260 >> $O/ts_layer.txt
done
cat $O/ts_layer.txt
