"""Development aid (GPU box): wall clock of resident encodes over a ladder of sizes (text; Default and Best), as a caller's context runs them."""
import os, sys, time, statistics
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "deflate-rs_amd")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import datagen, deflate_amd as da
ctx = da.Context(0)
big = datagen.text_like(33_000_000, 5)
for mb in (0.03, 0.06, 0.17, 0.25, 0.3, 0.5, 1, 1.5, 2, 2.2, 3, 4, 6, 8, 10, 16, 20, 32):
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
        print("%5.2f MB %-8s wall %.3f ms  %.0f MB/s" % (mb, lv.name, statistics.median(ws), n / statistics.median(ws) / 1e3))
