"""Development aid (GPU box): N device-resident encodes of a chosen input with the library MI355_DEFLATE_LIB names, nothing checked
-- for rocprofv3 over build variants.   loop_any.py <records96|records40|records256|text|dbrows|enwik|rows:WIDTH> <default|best|fast> [reps] [MB]"""
import os, sys
os.environ.setdefault("MI355_STAGE_CLOCKS", "1")  # (this aid reads the per-stage clocks: on for calls of every size)
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "deflate-rs_amd")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import datagen
import deflate_amd as da
kind, level = sys.argv[1], sys.argv[2]
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 3
mb = float(sys.argv[4]) if len(sys.argv) > 4 else 20.0
if kind.startswith("rows:"):  # rows of one length throughout: rows:40
    import numpy as np
    width = int(kind[5:])
    r = np.random.default_rng(width)
    rows = int(mb * 1e6) // width + 1
    a = np.tile(r.integers(0, 256, size=width, dtype=np.uint8), (rows, 1))
    a[:, 4:8] = np.arange(rows, dtype=np.uint32).view(np.uint8).reshape(rows, 4)
    cols = r.choice(np.arange(8, width), size=max(1, width // 8), replace=False)
    a[:, cols] = r.integers(0, 16, size=(rows, len(cols)), dtype=np.uint8)
    data = a.reshape(-1)[:int(mb * 1e6)].tobytes()
elif kind == "enwik":  # the bench's text
    data = datagen.text_like(int(mb * 1e6), 0x656E)
else:
    sil = datagen.silesia_like(scale=0.5)
    off = {"text": 0, "records96": 58.3e6, "records40": 109.5e6, "records256": 115.7e6, "dbrows": 137.3e6}[kind]
    data = sil[int(off * 0.5):int(off * 0.5) + int(mb * 1e6)]
lv = {"default": da.Compression.Default, "best": da.Compression.Best, "fast": da.Compression.Fast}[level]
ctx = da.Context(0)
n = len(data)
t = torch.frombuffer(bytearray(data), dtype=torch.uint8).cuda()
cap = da.bound(n) + 8
out = torch.empty(cap, dtype=torch.uint8, device="cuda")
for _ in range(reps):
    ctx.encode_device(t.data_ptr(), n, out.data_ptr(), cap, lv)
torch.cuda.synchronize()
print(kind, level, n, "bytes; match ms", round(ctx.info()["stage_ms"]["match"], 3))
