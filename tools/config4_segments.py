"""GPU box: BASELINE config 4 piece by piece -- the twelve pieces of the Silesia-like mix (tests/datagen.py silesia_like), each
encoded on its own at Compression::Best: match-stage ms per MB says which kind of data the walk is slow on.
    python tools/config4_segments.py > profiles/rNN_config4_segments.txt"""
import os, sys
os.environ.setdefault("MI355_STAGE_CLOCKS", "1")  # (this aid reads the per-stage clocks: on for calls of every size)
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in ("deflate-rs_amd", "tests"):
    sys.path.insert(0, os.path.join(ROOT, p))
import torch, datagen, deflate_amd as da
data = datagen.silesia_like(0x53494C45)
mb = lambda x: int(x * 1e6)
sizes = [10.2, 6.6, 41.5, 51.2, 6.2, 21.6, 33.6, 10.1, 10.0, 8.5, 7.3, 5.3]
kinds = ["text", "text", "text", "records of 96 B", "records of 40 B", "records of 256 B", "database rows", "database rows",
         "16-bit samples", "16-bit samples", "noise", "text"]
ctx = da.Context(0)
off = 0
print("%-3s %-18s %12s %9s %9s %9s %9s %8s" % ("#", "kind", "bytes", "total ms", "match ms", "ns/B mtch", "MB/s", "ratio"))
tot = 0.0
for i, (s, k) in enumerate(zip(sizes, kinds)):
    n = mb(s)
    piece = data[off:off + n]
    off += n
    t = torch.frombuffer(bytearray(piece), dtype=torch.uint8).cuda()
    cap = da.bound(n) + 8
    out = torch.empty(cap, dtype=torch.uint8, device="cuda")
    for _ in range(3):
        m = ctx.encode_device(t.data_ptr(), n, out.data_ptr(), cap, da.Compression.Best)
    info = ctx.info()
    tot += info["total_ms"]
    print("%-3d %-18s %12d %9.3f %9.3f %9.3f %9.0f %8.3f" % (i, k, n, info["total_ms"], info["stage_ms"]["match"], info["stage_ms"]["match"] * 1e6 / n,
                                                           n / info["total_ms"] / 1e3, m / n))
print("sum of the pieces %.2f ms for %d bytes = %.0f MB/s" % (tot, off, off / tot / 1e3))
