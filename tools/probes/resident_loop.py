"""Development aid (GPU box): N device-resident encodes of the 100 MB text with the library MI355_DEFLATE_LIB names, nothing
checked -- for rocprofv3 --kernel-trace --stats over builds whose output is not meant to be right (timing experiments)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "deflate-rs_amd")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import datagen
import deflate_amd as da
ctx = da.Context(0)
n = 100_000_000
t = torch.frombuffer(bytearray(datagen.text_like(n, 0xE8)), dtype=torch.uint8).cuda()
cap = da.bound(n) + 8
out = torch.empty(cap, dtype=torch.uint8, device="cuda")
for _ in range(int(sys.argv[1]) if len(sys.argv) > 1 else 5):
    try:
        ctx.encode_device(t.data_ptr(), n, out.data_ptr(), cap, da.Compression.Default)
    except Exception as e:  # (a build that leaves a stage out may well fail its own checks)
        print("encode:", e)
torch.cuda.synchronize()
