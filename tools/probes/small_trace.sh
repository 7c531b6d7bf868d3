#!/bin/bash
# development aid (GPU box): kernel trace of the last of ten pg11 encodes -- start, duration, gap to the one before
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/st && rocprofv3 --kernel-trace --output-format csv -d /tmp/st -- python $GRAFT_REPO_ROOT/tools/probes/small_trace.py 10 > /tmp/st.log 2>&1; true
python - <<'PY'
import csv, glob
f = glob.glob("/tmp/st/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in csv.DictReader(open(f))))
# the last encode: from the last k_sort on
last = max(i for i, r in enumerate(rows) if "k_sort" in r[2])
seq = rows[last - 2:]
t0 = seq[0][0]
prev = None
tot = 0
for s, e, name in seq:
    gap = (s - prev) / 1e3 if prev else 0.0
    print("%8.1f us  dur %7.1f  gap %6.1f  %s" % ((s - t0) / 1e3, (e - s) / 1e3, gap, name[:60]))
    prev = e
    tot += e - s
print("kernels %.1f us of %.1f us" % (tot / 1e3, (seq[-1][1] - t0) / 1e3))
PY
