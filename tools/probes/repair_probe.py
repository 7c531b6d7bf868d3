"""Development aid (GPU box): which small inputs make the speculative parse repair entries (spec_repaired) without failing
(spec_fallback)?  A fresh context per case."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "deflate-rs_amd")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import datagen, deflate_amd as da
text = datagen.text_like(1 << 20, 41)
noise = datagen.rng_bytes(1 << 20, 42)
cases = {
    "text + abcde x600": b"".join(text[i * 7000:(i + 1) * 7000] + bytes([97 + i % 5, 98, 99, 100 + i % 3, 101]) * 600 for i in range(60)),
    "text + zeros 1500": b"".join(text[i * 7000:(i + 1) * 7000] + bytes(1500) for i in range(60)),
    "text + zeros 3000": b"".join(text[i * 7000:(i + 1) * 7000] + bytes(3000) for i in range(60)),
    "text + period 300 x10": b"".join(text[i * 7000:(i + 1) * 7000] + noise[i * 300:(i + 1) * 300] * 10 for i in range(60)),
    "mixed 900K": datagen.mixed(900 * 1024, 44),
    "mixed 400K": datagen.mixed(400 * 1024, 45),
}
for name, data in cases.items():
    for lv in (da.Compression.Default, da.Compression.Fast, da.Compression.Best):
        c = da.Context(0)
        c.encode(data, lv)
        i = c.info()
        print("%-24s %-8s %8d bytes  repaired %4d  fallback %4d  passes %d" % (name, lv.name, len(data), i["spec_repaired"], i["spec_fallback"], i["passes"]))
        c.close()
print("-- the test's data, five runs each")
text2 = datagen.text_like(1100 * 1024, 41)
for gap in (bytes(1500), bytes(3000)):
    data = b"".join(text2[i * 7000:(i + 1) * 7000] + gap for i in range(60))
    for lv in (da.Compression.Default, da.Compression.Fast, da.Compression.Best):
        r = []
        for _ in range(5):
            c = da.Context(0)
            c.encode(data, lv)
            i = c.info()
            r.append((i["spec_repaired"], i["spec_fallback"]))
            c.close()
        print(len(gap), lv.name, r)
print("-- explicit options / compat")
data = b"".join(text2[i * 7000:(i + 1) * 7000] + bytes(3000) for i in range(60))
for opts, compat in ((da.CompressionOptions(128, 32, 1), 0), (da.CompressionOptions(128, 32, 1), 1), (da.Compression.Default, 1)):
    c = da.Context(0)
    c.encode(data, opts, compat=compat)
    i = c.info()
    print(opts if not isinstance(opts, da.CompressionOptions) else "CompressionOptions(128, 32, 1)", compat, i["spec_repaired"], i["spec_fallback"], i["n_blocks"], i["q13_hits"])
    c.close()
