# HBM traffic (FETCH_SIZE) of the prefill DMA GEMM (M=798) and the streaming kernel (M=16) on the gate_up shape.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/gemm_fetch_size.txt
rm -f $OUT
for M in 798 16; do
  timeout 120 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d /tmp/fs$M -o p -- python $R/tools/gemm_prof.py gate_up $M planes > /tmp/fs$M.log 2>&1
  db=$(find /tmp/fs$M -name "*.db" | head -1)
  echo "## rocprofv3 --kernel-trace --pmc FETCH_SIZE -- python tools/gemm_prof.py gate_up $M planes   (W 283.1 MB, 4 different weight sets in rotation)" >> $OUT
  if [ -n "$db" ]; then python $R/tools/prof_db.py $db | grep -v "fill_hash\|^$" | grep -i "gemm\|calls\|PMC\|FETCH" >> $OUT; else tail -3 /tmp/fs$M.log >> $OUT; fi
done
