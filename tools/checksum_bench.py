"""GPU box: time the checksum kernels (Adler-32 of zlib, CRC-32 of gzip) on a resident buffer."""
import os, sys, time, zlib
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "deflate-rs_amd")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import datagen, deflate_amd as da
n = int(sys.argv[1]) if len(sys.argv) > 1 else 100_000_000
data = datagen.text_like(n, 0x656E)
t = torch.frombuffer(bytearray(data), dtype=torch.uint8).cuda()
torch.cuda.synchronize()
ctx = da.Context(0)
for name, fn, ref in (("crc32", ctx.crc32_device, zlib.crc32(data)), ("adler32", ctx.adler32_device, zlib.adler32(data))):
    assert fn(t.data_ptr(), n) == ref
    best = 1e9
    for _ in range(5):
        t0 = time.perf_counter(); fn(t.data_ptr(), n); best = min(best, time.perf_counter() - t0)
    print("%s: %.3f ms wall per call (incl. launch + result copy) = %.1f GB/s" % (name, best * 1e3, n / best / 1e9))
