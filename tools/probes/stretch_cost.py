"""Development aid (GPU box): what a few short periodic stretches cost a large call -- 100 MB of text with a zero run of 2 KB every
megabyte -- for the library named by MI355_DEFLATE_LIB (default: the product)."""
import os, sys, time
os.environ.setdefault("MI355_STAGE_CLOCKS", "1")
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "deflate-rs_amd")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import datagen, deflate_amd as da
base = datagen.text_like(100_000_000, 0x656E)
for name, data in (("text", base), ("text + 2 KB of zeros every MB", b"".join(base[i:i + 998_000] + bytes(2000) for i in range(0, 100_000_000, 1_000_000)))):
    n = len(data)
    t = torch.frombuffer(bytearray(data), dtype=torch.uint8).cuda()
    cap = da.bound(n) + 8
    out = torch.empty(cap, dtype=torch.uint8, device="cuda")
    ctx = da.Context(0)
    ws = []
    for k in range(6):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        ctx.encode_device(t.data_ptr(), n, out.data_ptr(), cap)
        ws.append((time.perf_counter() - t0) * 1e3)
    i = ctx.info()
    print("%-34s calls (ms): %s   repaired %d fallback %d  parse stage %.3f ms" % (name, " ".join("%.2f" % w for w in ws), i["spec_repaired"], i["spec_fallback"], i["stage_ms"]["parse"]))
    ctx.close()
