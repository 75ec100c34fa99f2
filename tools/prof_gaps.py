"""Idle gaps on the GPU timeline of a rocprofv3 kernel trace: for each kernel, the average time between the end of
the previous dispatch and its own start (same process, ordered by start time).
    python tools/prof_gaps.py <results.db>"""
import collections
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
rows = db.execute("select name, start, end from kernels order by start").fetchall()
gap = collections.defaultdict(list)
prev_end = None
for name, s, e in rows:
    if prev_end is not None:
        gap[name.split("(")[0][-60:]].append((s - prev_end) / 1000.0)
    prev_end = max(prev_end or 0, e)
print(f"{'calls':>6} {'avg_gap_us':>10} {'median':>8} {'total_ms':>9}  kernel (gap BEFORE it)")
for k, v in sorted(gap.items(), key=lambda kv: -sum(kv[1])):
    if len(v) < 20:
        continue
    v2 = sorted(v)
    print(f"{len(v):6d} {sum(v) / len(v):10.2f} {v2[len(v2) // 2]:8.2f} {sum(v) / 1000:9.2f}  {k}")
