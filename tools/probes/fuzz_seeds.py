"""development aid (GPU box): run seeds of tools/fuzz_gpu.py and print the ones that differ"""
import os, sys
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
for p in ("deflate-rs_amd", "tests", "tools"):
    sys.path.insert(0, os.path.join(ROOT, p))
import torch  # noqa
import deflate_amd as da
import fuzz_gpu
ctx = da.Context(0)
lo, hi = int(sys.argv[1]), int(sys.argv[2])
bad = 0
for seed in range(lo, hi):
    r = fuzz_gpu.one(seed, ctx)
    if r.startswith("DIFF"):
        bad += 1
        print("seed", seed, r[:600])
print("seeds", lo, hi, "differing", bad)
