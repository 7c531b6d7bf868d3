"""development aid (GPU box): the speculative parse's counters over repeated encodes of the Silesia-like mix and the text --
what was repaired and whether a call fell back to the exact parse must not depend on the run"""
import os, sys
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
for p in ("deflate-rs_amd", "tests"):
    sys.path.insert(0, os.path.join(ROOT, p))
import torch, datagen, deflate_amd as da
sys.path.insert(0, ROOT)
import bench
for name, data, lvl in (("silesia", bench.make_input("silesia", 212_100_000, 0), da.Compression.Best),
                        ("text", datagen.text_like(50_000_000, 0x656E), da.Compression.Default)):
    if data is None:
        continue
    ctx = da.Context(0)
    t = torch.frombuffer(bytearray(data), dtype=torch.uint8).cuda()
    cap = da.bound(len(data)) + 8
    out = torch.empty(cap, dtype=torch.uint8, device="cuda")
    seen = []
    for _ in range(6):
        n = ctx.encode_device(t.data_ptr(), len(data), out.data_ptr(), cap, lvl)
        i = ctx.info()
        seen.append((i["spec_repaired"], i["spec_fallback"], n))
    print(name, seen)
    ctx.close()
