#!/bin/bash
# round 5, call G: tensor-parallel tests after the wider bulk all-reduce, one rank of TP = W (loop-back) on the final code, its kernel
# trace, a stage-marked trace of the bench, the other workloads' bench lines
set -x
O=gpurun_out/r5_g; mkdir -p $O
cd /tmp && export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0 && cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_tp_p2p.py tests/test_gpu_tp_multiprocess.py tests/test_gpu_tp_shards.py tests/test_gpu_parity_real_size.py -q -m gpu --durations=8 > $O/pytest_tp.txt 2>&1
tail -14 $O/pytest_tp.txt
timeout 300 python -m pytest tests/test_gpu_kernels.py -q -m gpu -k "attention or ring or gemv_ksplit" > $O/pytest_attn.txt 2>&1
tail -4 $O/pytest_attn.txt
timeout 900 python tools/tp_shard_step.py --worlds 1,2,4,8 --out $O/r5_tp_shard_step.json > $O/tp.log 2>&1
tail -3 $O/tp.log
rm -rf /tmp/kt8
timeout 300 rocprofv3 --kernel-trace -d /tmp/kt8 -o p -- python tools/tp_shard_step.py --worlds 8 --steps 8 --out $O/tp8_traced.json > $O/tp8_trace.log 2>&1
(echo "## rocprofv3 --kernel-trace -- python tools/tp_shard_step.py --worlds 8 --steps 8   (one rank of TP = 8, loop-back exchange, MI355X, round 5)"; python tools/prof_db.py $(find /tmp/kt8 -name "*.db" | head -1)) > $O/r5_tp8_shard_kernel_trace.txt
head -24 $O/r5_tp8_shard_kernel_trace.txt
rm -rf /tmp/ktm
timeout 300 rocprofv3 --kernel-trace --marker-trace -d /tmp/ktm -o p -- python bench.py --steps 4 --warmup 2 --no-cpu-baseline --ttft-runs 2 > $O/bench_marker.log 2>&1
db=$(find /tmp/ktm -name "*.db" | head -1)
python - "$db" > $O/marker_tables.txt 2>&1 <<'PY'
import sqlite3, sys
db = sqlite3.connect(sys.argv[1]); cur = db.cursor()
names = [r[0] for r in cur.execute("select name from sqlite_master where type in ('table','view') order by name")]
print(names)
for t in names:
    if any(k in t.lower() for k in ("region", "marker", "roctx")):
        try:
            cols = [c[1] for c in cur.execute(f"pragma table_info('{t}')")]
            n = cur.execute(f"select count(*) from '{t}'").fetchone()[0]
            print(t, n, cols)
            for row in cur.execute(f"select * from '{t}' limit 5"):
                print("   ", row)
        except Exception as e:
            print(t, "error", e)
PY
head -30 $O/marker_tables.txt
timeout 400 python bench.py --series 30 --lengths mixed --steps 32 --warmup 8 --no-cpu-baseline > $O/r5_bench_cfg4_30xmixed.json 2> $O/cfg4.err; tail -c 600 $O/r5_bench_cfg4_30xmixed.json
timeout 400 python bench.py --batch 16 --weights fp8 --series 8 --length 1024 --steps 24 --warmup 6 --no-cpu-baseline > $O/r5_bench_cfg5_fp8_8x1024_b16.json 2> $O/cfg5.err; tail -c 600 $O/r5_bench_cfg5_fp8_8x1024_b16.json
timeout 300 python bench.py --model chatts-8b --series 1 --length 256 --steps 32 --warmup 8 --no-cpu-baseline > $O/r5_bench_8b_cfg2.json 2> $O/8b.err; tail -c 400 $O/r5_bench_8b_cfg2.json
