"""Development aid (GPU box): wall clock of resident encodes between 2 and 5 MB, text and noise, for the library MI355_DEFLATE_LIB names."""
import os, sys, time, statistics
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "deflate-rs_amd")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import datagen, deflate_amd as da
ctx = da.Context(0)
src = {"text": datagen.text_like(5_000_000, 5), "noise": datagen.rng_bytes(5_000_000, 6)}
for kind in ("text", "noise"):
    for mb in (2.0, 2.2, 3.0, 4.0, 4.3):
        n = int(mb * 1e6)
        t = torch.frombuffer(bytearray(src[kind][:n]), dtype=torch.uint8).cuda()
        cap = da.bound(n) + 8
        out = torch.empty(cap, dtype=torch.uint8, device="cuda")
        for _ in range(3):
            ctx.encode_device(t.data_ptr(), n, out.data_ptr(), cap)
        ws = []
        for _ in range(20):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            ctx.encode_device(t.data_ptr(), n, out.data_ptr(), cap)
            ws.append((time.perf_counter() - t0) * 1e3)
        print("%-6s %4.1f MB  wall %.3f ms" % (kind, mb, statistics.median(ws)))
