#!/bin/bash
# round 5: where the waves of the batch-1 decode step (headline bench) spend their cycles - one SQ counter pass
# (WAVE_CYCLES = WAIT_ANY [parked at s_waitcnt / barrier] + WAIT_INST_ANY [issue stall] + ACTIVE_INST_ANY, quad-cycles; MI355X_MICROARCH.md)
R="${GRAFT_REPO_ROOT:-.}"; O=$R/gpurun_out/r5_q; mkdir -p $O
cd /tmp && export TMPDIR=/tmp; rm -rf /tmp/sqh
C5="python bench.py --steps 8 --warmup 2 --no-cpu-baseline --ttft-runs 1"
timeout 500 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU -d /tmp/sqh -o p -- python $R/bench.py --steps 8 --warmup 2 --no-cpu-baseline --ttft-runs 1 > /tmp/sqh.log 2>&1
db=$(find /tmp/sqh -name "*.db" | head -1)
(echo "## rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU -- $C5   (MI355X, round 5, final code)"
 python - $db <<'PY'
import sqlite3, sys
cur = sqlite3.connect(sys.argv[1]).cursor()
rows = cur.execute("""select kernel_name, grid_size, counter_name, count(*), avg(value), avg(duration)/1000.0 from counters_collection
                      where kernel_name like '%chatts%' and kernel_name not like '%fill_hash%' group by kernel_name, grid_size, counter_name""").fetchall()
k = {}
for name, grid, c, n, v, d in rows:
    e = k.setdefault((name, grid), {"n": n, "us": d}); e[c] = v
print(f"{'calls':>6} {'avg_us':>8} {'grid':>9} {'waves':>7} {'parked':>7} {'stall':>6} {'active':>6} {'valu':>6}  kernel   (shares of SQ_WAVE_CYCLES)")
for (name, grid), e in sorted(k.items(), key=lambda kv: -kv[1]["us"] * kv[1]["n"])[:24]:
    wc = e.get("SQ_WAVE_CYCLES") or 1
    print(f"{e['n']:6d} {e['us']:8.2f} {grid:9d} {e.get('SQ_WAVES', 0):7.0f} {e.get('SQ_WAIT_ANY', 0) / wc:7.2f} {e.get('SQ_WAIT_INST_ANY', 0) / wc:6.2f} "
          f"{e.get('SQ_ACTIVE_INST_ANY', 0) / wc:6.2f} {e.get('SQ_ACTIVE_INST_VALU', 0) / wc:6.2f}  {name[:90]}")
PY
) > $O/r5_bench_pmc_sq.txt 2>&1
head -30 $O/r5_bench_pmc_sq.txt | cut -c1-200; tail -3 /tmp/sqh.log | cut -c1-200
