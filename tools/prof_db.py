"""Summarise a rocprofv3 results .db (rocpd sqlite): per-kernel, per-launch-shape call count / avg / min duration,
and (when a --pmc pass was recorded) the average counter value per kernel shape.
    python tools/prof_db.py <results.db> [--all]"""
import sqlite3
import sys


def main():
    db = sqlite3.connect(sys.argv[1])
    cur = db.cursor()
    only = "" if "--all" in sys.argv else "where name like '%chatts%'"
    print(f"# kernel-trace summary of {sys.argv[1]}")
    print(f"{'calls':>6} {'avg_us':>9} {'min_us':>9} {'total_ms':>9} {'grid':>9} {'wg':>5} {'lds':>6} {'vgpr':>4}  kernel")
    q = f"""select name, grid_x*grid_y*grid_z, workgroup_x, count(*), avg(end-start)/1000.0, min(end-start)/1000.0,
                   sum(end-start)/1e6, lds_size, vgpr_count from kernels {only}
            group by name, grid_x, grid_y, grid_z order by sum(end-start) desc"""
    for name, grid, wg, n, avg, mn, tot, lds, vgpr in cur.execute(q):
        print(f"{n:6d} {avg:9.2f} {mn:9.2f} {tot:9.2f} {grid:9d} {wg:5d} {lds:6d} {vgpr:4d}  {name[:110]}")
    try:
        rows = cur.execute("""select kernel_name, grid_size, workgroup_size, counter_name, count(*), avg(value), avg(duration)/1000.0
                              from counters_collection where kernel_name like '%chatts%'
                              group by kernel_name, grid_size, counter_name order by avg(value) desc""").fetchall()
    except sqlite3.Error:
        rows = []
    if rows:
        print("\n# PMC pass (values as rocprofv3 reports them; FETCH_SIZE is in KiB and, on gfx950, counts 64 B per 128 B\n"
              "# request for wide coalesced streams -> multiply by 2 for bytes, MI355X_MICROARCH.md section HBM)")
        print(f"{'calls':>6} {'avg_value':>14} {'avg_us':>9} {'grid':>9} {'wg':>5}  counter     kernel")
        for name, grid, wg, cname, n, val, dur in rows:
            print(f"{n:6d} {val:14.1f} {dur:9.2f} {grid:9d} {wg:5d}  {cname:10s}  {name[:100]}")


def regions(path):
    """roctx stage ranges (chatts.* - csrc/api.hip stage_push / stage_pop) of a --marker-trace run: host-side ENQUEUE time per stage"""
    import json
    cur = sqlite3.connect(path).cursor()
    try:
        rows = cur.execute("select extdata, duration from regions where category like 'MARKER%'").fetchall()
    except sqlite3.Error:
        return
    agg = {}
    for ext, dur in rows:
        try:
            name = json.loads(ext).get("message", "?")
        except Exception:
            name = "?"
        a = agg.setdefault(name, [0, 0.0])
        a[0] += 1
        a[1] += dur / 1000.0
    if agg:
        print("\n# roctx stage ranges (host-side enqueue time; names from csrc: chatts.<stage>)")
        print(f"{'calls':>6} {'avg_us':>10} {'total_ms':>9}  range")
        for name, (n, tot) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
            print(f"{n:6d} {tot / n:10.1f} {tot / 1000.0:9.2f}  {name}")


if __name__ == "__main__":
    main()
    regions(sys.argv[1])
