#!/bin/bash
# development aid (GPU box): the kernel sequence of ONE resident encode of the bench's 100 MB text (the last of three): start, duration, gap
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/bt && rocprofv3 --kernel-trace --output-format csv -d /tmp/bt -- python $GRAFT_REPO_ROOT/tools/probes/loop_any.py enwik default 3 100 > /tmp/bt.log 2>&1; true
python - <<'PY'
import csv, glob
f = glob.glob("/tmp/bt/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in csv.DictReader(open(f))))
last = max(i for i, r in enumerate(rows) if "k_sort" in r[2])
seq = rows[last - 3:]
t0 = seq[0][0]
prev = None
for s, e, name in seq:
    gap = (s - prev) / 1e3 if prev else 0.0
    print("%9.1f us  dur %8.1f  gap %7.1f  %s" % ((s - t0) / 1e3, (e - s) / 1e3, gap, name[:70]))
    prev = e
PY
