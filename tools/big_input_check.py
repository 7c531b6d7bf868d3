import os
os.environ.setdefault("MI355_STAGE_CLOCKS", "1")  # (this aid reads the per-stage clocks: on for calls of every size)
import sys, zlib, time
sys.path.insert(0,"deflate-rs_amd"); sys.path.insert(0,"tests")
import torch, datagen, deflate_amd as da
ctx = da.Context(0)
n = 1_500_000_000
t0=time.time(); base = datagen.text_like(100_000_000, 77); data = (base * 15)[:n]; print("gen", time.time()-t0, flush=True)
t = torch.frombuffer(bytearray(data), dtype=torch.uint8).cuda()
cap = da.bound(n) + 64
out = torch.empty(cap, dtype=torch.uint8, device="cuda"); torch.cuda.synchronize()
for lvl in (da.Compression.Default, da.Compression.Fast):
    t0=time.time(); m = ctx.encode_device(t.data_ptr(), n, out.data_ptr(), cap, lvl); torch.cuda.synchronize(); dt=time.time()-t0
    z = bytes(out[:m].cpu().numpy()); info = ctx.info()
    d = zlib.decompressobj(-15); got = d.decompress(z); ok = (got == data) and d.eof
    print(lvl, "out", m, "wall %.3f s"%dt, "gpu %.1f ms"%info["total_ms"], "%.2f GB/s"%(n/info["total_ms"]/1e6), "roundtrip", ok, flush=True)
    del got
