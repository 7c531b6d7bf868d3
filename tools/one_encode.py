"""Development aid (GPU box): one resident encode of pg11 (or N bytes of text), for debug builds that print."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "deflate-rs_amd")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch, datagen, deflate_amd as da
ctx = da.Context(0)
data = open(os.path.join(ROOT, "tests/golden/ref_inputs/pg11.txt"), "rb").read() if len(sys.argv) < 2 else datagen.text_like(int(sys.argv[1]), 2)
t = torch.frombuffer(bytearray(data), dtype=torch.uint8).cuda()
cap = da.bound(len(data)) + 8
out = torch.empty(cap, dtype=torch.uint8, device="cuda")
for _ in range(3):
    ctx.encode_device(t.data_ptr(), len(data), out.data_ptr(), cap, da.Compression.Default)
torch.cuda.synchronize()
print(ctx.info()["stage_ms"])
