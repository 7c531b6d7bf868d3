#!/bin/bash
# development aid (GPU box): kernel timeline of one small resident encode (pg11, Default) under rocprofv3
cd /tmp && export TMPDIR=/tmp
cat > /tmp/tl_driver.py <<PY
import os, sys
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, os.path.join(ROOT, "deflate-rs_amd")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch, datagen, deflate_amd as da
ctx = da.Context(0)
kind = sys.argv[1] if len(sys.argv) > 1 else "pg11"
data = open(os.path.join(ROOT, "tests/golden/ref_inputs/pg11.txt"), "rb").read() if kind == "pg11" else datagen.text_like(2_000_000, 2)
t = torch.frombuffer(bytearray(data), dtype=torch.uint8).cuda()
cap = da.bound(len(data)) + 8
out = torch.empty(cap, dtype=torch.uint8, device="cuda")
for _ in range(10):
    ctx.encode_device(t.data_ptr(), len(data), out.data_ptr(), cap, da.Compression.Default)
torch.cuda.synchronize()
PY
rm -rf /tmp/tl && rocprofv3 --kernel-trace -d /tmp/tl -o tl --output-format csv -- python /tmp/tl_driver.py "$@" > /tmp/tl.log 2>&1
python - <<'PY'
import csv, glob
fs = glob.glob("/tmp/tl/**/*kernel_trace.csv", recursive=True)
if not fs:
    print(open("/tmp/tl.log").read()[-2000:]); raise SystemExit
rows = sorted(csv.DictReader(open(fs[0])), key=lambda r: int(r["Start_Timestamp"]))
starts = [i for i, r in enumerate(rows) if "k_sort" in r["Kernel_Name"]]
i0, i1 = starts[-2], starts[-1]
t0 = int(rows[i0]["Start_Timestamp"]); prev_end = t0
for r in rows[i0:i1]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    print("%-30s start %7.1f us  dur %6.1f us  gap before %5.1f us" % (r["Kernel_Name"].split("(")[0][-30:], (s - t0) / 1e3, (e - s) / 1e3, (s - prev_end) / 1e3))
    prev_end = e
print("span %.1f us, kernels %d, sum of durations %.1f us" % ((prev_end - t0) / 1e3, i1 - i0, sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in rows[i0:i1]) / 1e3))
PY
