"""Development aid (GPU box): N resident encodes of pg11.txt -- run under rocprofv3 --kernel-trace to see a small input's
launch sequence (tools/probes/small_trace.sh prints kernel times and the gaps between them)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "deflate-rs_amd")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import deflate_amd as da
ctx = da.Context(0)
data = open(os.path.join(ROOT, "tests/golden/ref_inputs/pg11.txt"), "rb").read()
n = len(data)
t = torch.frombuffer(bytearray(data), dtype=torch.uint8).cuda()
cap = da.bound(n) + 8
out = torch.empty(cap, dtype=torch.uint8, device="cuda")
for _ in range(int(sys.argv[1]) if len(sys.argv) > 1 else 10):
    ctx.encode_device(t.data_ptr(), n, out.data_ptr(), cap, da.Compression.Default)
torch.cuda.synchronize()
