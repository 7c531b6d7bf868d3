"""Reads a rocprofv3 results database (--kernel-trace) of calls of the host API and prints, per call, the chain of sorts and walks
relative to the call's first kernel, and when its last pack ended: where a call waits for its input shows as a gap in the chain.
    python tools/probes/chain.py <results.db>"""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
cur = db.cursor()
rows = list(cur.execute("select name,start,end,grid_x,workgroup_x from kernels order by start"))
calls = [i for i, r in enumerate(rows) if "k_sort" in r[0] and r[3] // r[4] == 64]
for j, i0 in enumerate(calls):
    i1 = calls[j + 1] if j + 1 < len(calls) else len(rows)
    seg = rows[i0:i1]
    t0 = seg[0][1]
    out = []
    for r in seg:
        if "k_match3" in r[0] or "k_sort" in r[0]:
            out.append("%s@%.3f+%.0f" % ("M" if "match3" in r[0] else "S", (r[1] - t0) / 1e6, (r[2] - r[1]) / 1e3))
    packs = [r[2] for r in seg if "k_pack" in r[0]]
    print(j, " ".join(out), "| last pack end %.3f" % ((max(packs) - t0) / 1e6) if packs else "")
