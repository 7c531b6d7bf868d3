"""development aid (GPU box): mi355_deflate_encode_multi on page-locked host buffers, N ranks on device 0"""
import os, sys, time
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
for p in ("deflate-rs_amd", "tests"):
    sys.path.insert(0, os.path.join(ROOT, p))
import torch, datagen, deflate_amd as da, oracle_binding as ob
n_ranks = int(sys.argv[1]) if len(sys.argv) > 1 else 2
data = datagen.text_like(100_000_000 * n_ranks, 0x77)
h_in = torch.frombuffer(bytearray(data), dtype=torch.uint8).pin_memory()
cap = da.bound(len(data)) + 64
h_out = torch.empty(cap, dtype=torch.uint8).pin_memory()
m = da.MultiGpu([0] * n_ranks)
ts = []
for _ in range(6):
    t0 = time.perf_counter()
    n = m.encode_host_ptr(h_in.data_ptr(), len(data), h_out.data_ptr(), cap, da.Compression.Default)
    ts.append((time.perf_counter() - t0) * 1e3)
want = ob.encode(data, level=ob.DEFAULT)
print(n_ranks, "ranks, host buffers:", ["%.2f" % t for t in ts], "ms; %.0f MB/s; same as oracle:" % (len(data) / min(ts) / 1e3), bytes(h_out[:n].numpy()) == want, m.trace())
