#!/bin/bash
# development aid (GPU box): copies and kernels of one host-buffer call (100 MB, pinned) on a time line
cd /tmp && export TMPDIR=/tmp
cat > /tmp/ha_driver.py <<PY
import os, sys, ctypes as C
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, os.path.join(ROOT, "deflate-rs_amd")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch, datagen, deflate_amd as da
ctx = da.Context(0)
data = datagen.text_like(100_000_000, 2)
hin = torch.frombuffer(bytearray(data), dtype=torch.uint8).pin_memory()
cap = da.bound(len(data)) + 64
hout = torch.empty(cap, dtype=torch.uint8).pin_memory()
for _ in range(4):
    n = ctx.encode_host_ptr(hin.data_ptr(), len(data), hout.data_ptr(), cap, da.Compression.Default)
print(n)
PY
rm -rf /tmp/ha && rocprofv3 --kernel-trace --memory-copy-trace -d /tmp/ha -o ha --output-format csv -- python /tmp/ha_driver.py > /tmp/ha.log 2>&1
python - <<'PY'
import csv, glob
ev = []
for f in glob.glob("/tmp/ha/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0][-26:]))
for f in glob.glob("/tmp/ha/**/*memory_copy_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "COPY " + r.get("Direction", r.get("Kind", "?")) + " " + r.get("Bytes", "")))
if not ev:
    print(open("/tmp/ha.log").read()[-1500:]); raise SystemExit
ev.sort()
sorts = [i for i, e in enumerate(ev) if "k_sort" in e[2]]
# the last call: from the first big H2D copy before the last group of k_sort launches
i1 = len(ev)
i0 = max(i for i, e in enumerate(ev[:sorts[-8] if len(sorts) >= 8 else sorts[0]]) if e[2].startswith("COPY") ) - 8
i0 = max(i0, 0)
t0 = ev[i0][0]
for s, e, nm in ev[i0:i1]:
    if e - s > 20000 or nm.startswith("COPY"):
        print("%-40s %9.1f .. %9.1f us (%7.1f)" % (nm, (s - t0) / 1e3, (e - t0) / 1e3, (e - s) / 1e3))
PY
