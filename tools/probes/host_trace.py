"""Development aid (GPU box): N host calls (pinned buffers) of the 100 MB text -- run under rocprofv3 --kernel-trace
--memory-copy-trace to see the streamed call's timeline (tools/probes/host_trace.sh prints the last call)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "deflate-rs_amd")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import time
import torch
import datagen
import deflate_amd as da
ctx = da.Context(0)
n = 100_000_000
data = datagen.text_like(n, 0xE8)
hin = torch.frombuffer(bytearray(data), dtype=torch.uint8).pin_memory()
cap = da.bound(n) + 8
hout = torch.empty(cap, dtype=torch.uint8).pin_memory()
for _ in range(int(sys.argv[1]) if len(sys.argv) > 1 else 5):
    t0 = time.time()
    k = ctx.encode_host_ptr(hin.data_ptr(), n, hout.data_ptr(), cap, da.Compression.Default)
    print("call ms", round((time.time() - t0) * 1e3, 3), k, flush=True)
