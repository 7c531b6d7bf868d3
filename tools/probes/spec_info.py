import os, sys
os.environ.setdefault("MI355_STAGE_CLOCKS", "1")  # (this aid reads the per-stage clocks: on for calls of every size)
ROOT="/root/repo"
sys.path.insert(0, os.path.join(ROOT, "deflate-rs_amd")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch, datagen, deflate_amd as da
ctx = da.Context(0)
data = datagen.text_like(100_000_000, 0x656E)
t = torch.frombuffer(bytearray(data), dtype=torch.uint8).cuda()
cap = da.bound(len(data)) + 8
out = torch.empty(cap, dtype=torch.uint8, device="cuda")
for _ in range(3):
    ctx.encode_device(t.data_ptr(), len(data), out.data_ptr(), cap, da.Compression.Default)
i = ctx.info()
print(os.environ.get("MI355_DEFLATE_LIB","product"), {k: i[k] for k in i if "spec" in k}, i["stage_ms"])
