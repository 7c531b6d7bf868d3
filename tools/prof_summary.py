"""Turn a rocprofv3 results .db (rocpd sqlite) into the per-kernel stats table committed under
profiles/.  usage: python tools/prof_summary.py gpurun_out/prof1/bench_results.db > profiles/x.txt"""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
cur = db.cursor()
rows = cur.execute(
    "select name, count(*), sum(end-start)/1e6, avg(end-start)/1e3, min(end-start)/1e3, max(end-start)/1e3, "
    "max(vgpr_count), max(lds_size), max(scratch_size), max(grid_x), max(workgroup_x) "
    "from kernels group by name order by 3 desc").fetchall()
tot = sum(r[2] for r in rows)
print("# rocprofv3 --kernel-trace --stats summary (from %s)" % sys.argv[1])
if len(sys.argv) > 2:
    print("# command: " + " ".join(sys.argv[2:]))
print("%-44s %5s %10s %10s %10s %10s %6s %5s %7s %9s %5s" % (
    "kernel", "calls", "total_ms", "avg_us", "min_us", "max_us", "pct", "vgpr", "lds", "grid", "wg"))
for r in rows:
    name = r[0].split("(")[0].replace("mi355::", "")
    print("%-44s %5d %10.3f %10.1f %10.1f %10.1f %6.1f %5d %7d %9d %5d" % (
        name[:44], r[1], r[2], r[3], r[4], r[5], 100 * r[2] / tot, r[6] or 0, r[7] or 0, r[9] or 0, r[10] or 0))
print("# total kernel time %.3f ms" % tot)
