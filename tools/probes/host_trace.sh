#!/bin/bash
# development aid (GPU box): kernel + copy trace of the last of five host calls of the 100 MB text
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/ht && rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d /tmp/ht -- python $GRAFT_REPO_ROOT/tools/probes/host_trace.py 5 > /tmp/ht.log 2>&1; grep "call ms" /tmp/ht.log
python - <<'PY'
import csv, glob
rows = []
for f in glob.glob("/tmp/ht/**/*kernel_trace.csv", recursive=True):
    rows += [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0][-28:]) for r in csv.DictReader(open(f))]
for f in glob.glob("/tmp/ht/**/*memory_copy_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "COPY " + r.get("Direction", r.get("Name", "?"))[-24:]))
rows.sort()
# the last call: from the last k_sort whose predecessor k_sort is > 3 ms earlier
sorts = [i for i, r in enumerate(rows) if "k_sort" in r[2]]
first = sorts[-1]
for a, b in zip(sorts, sorts[1:]):
    if rows[b][0] - rows[a][0] > 3_000_000: first = b
# include copies shortly before
t0 = rows[first][0]
seq = [r for r in rows if r[0] >= t0 - 400_000]
t0 = seq[0][0]
for s, e, name in seq:
    if (e - s) > 15_000 or "COPY" in name:
        print("%8.1f us  dur %7.1f  %s" % ((s - t0) / 1e3, (e - s) / 1e3, name))
print("span %.1f us" % ((max(e for s, e, n in seq) - t0) / 1e3))
PY
