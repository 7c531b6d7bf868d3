"""Development aid (GPU box): stage times of resident encodes over a ladder of sizes (text, Default)."""
import os, sys, time, statistics
os.environ.setdefault("MI355_STAGE_CLOCKS", "1")  # (this aid reads the per-stage clocks: on for calls of every size)
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "deflate-rs_amd")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import datagen, deflate_amd as da
ctx = da.Context(0)
big = datagen.text_like(33_000_000, 5)
for mb in (0.17, 0.5, 1, 2, 3, 4, 6, 8, 10, 16, 20, 32):
    n = int(mb * 1e6)
    t = torch.frombuffer(bytearray(big[:n]), dtype=torch.uint8).cuda()
    cap = da.bound(n) + 8
    out = torch.empty(cap, dtype=torch.uint8, device="cuda")
    for lv in (da.Compression.Default, da.Compression.Best):
        for _ in range(3):
            ctx.encode_device(t.data_ptr(), n, out.data_ptr(), cap, lv)
        ws = []
        for _ in range(10):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            ctx.encode_device(t.data_ptr(), n, out.data_ptr(), cap, lv)
            ws.append((time.perf_counter() - t0) * 1e3)
        i = ctx.info()
        print("%5.2f MB %-8s wall %.3f ms  match %.3f" % (mb, lv.name, statistics.median(ws), i["stage_ms"]["match"]))
