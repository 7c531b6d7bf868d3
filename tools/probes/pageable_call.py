"""GPU box: the drop-in call on pageable memory (mi355_deflate_encode, numpy buffers) -- ms per call by thread count, beside the
page-locked call and the runtime's own copies.  MI355_BOUNCE_TRACE=1 prints every session's timeline on stderr.
    python tools/probes/pageable_call.py [--threads 2,4,8,12,16] [--reps 20] [--size 100000000]"""
import argparse
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in ("deflate-rs_amd", "tests"):
    sys.path.insert(0, os.path.join(ROOT, p))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--threads", default="2,4,8,12,16")
    ap.add_argument("--reps", type=int, default=20)
    ap.add_argument("--size", type=int, default=100_000_000)
    ap.add_argument("--level", default="default")
    args = ap.parse_args()
    import numpy as np
    import torch
    import datagen
    import deflate_amd as da
    data = datagen.text_like(args.size, 0x656E77696B38)
    opts = {"default": da.CompressionOptions.default, "best": da.CompressionOptions.high, "fast": da.CompressionOptions.fast}[args.level]()
    cap = da.bound(args.size) + 64
    p_in = np.frombuffer(data, dtype=np.uint8).copy()
    p_out = np.zeros(cap, dtype=np.uint8)
    h_in = torch.frombuffer(bytearray(data), dtype=torch.uint8).pin_memory()
    h_out = torch.empty(cap, dtype=torch.uint8).pin_memory()

    def run(ctx, i, o):
        n = ctx.encode_host_ptr(i, args.size, o, cap, opts)
        ts = []
        for _ in range(args.reps):
            t = time.perf_counter()
            n = ctx.encode_host_ptr(i, args.size, o, cap, opts)
            ts.append((time.perf_counter() - t) * 1e3)
        ts.sort()
        return n, ts[0], ts[len(ts) // 2], ts[-1], sum(ts) / len(ts)
    ctx = da.Context(0)
    ctx.reserve(args.size, host_api=True)
    n, lo, med, hi, mean = run(ctx, h_in.data_ptr(), h_out.data_ptr())
    want = bytes(h_out[:n].numpy())
    print("page-locked            : min %.3f median %.3f max %.3f mean %.3f ms  (%.0f MB/s by the mean)" % (lo, med, hi, mean, args.size / mean / 1e3))
    ctx.config(da.Context.CFG_HOST_BOUNCE, 0)
    n, lo, med, hi, mean = run(ctx, p_in.ctypes.data, p_out.ctypes.data)
    assert bytes(p_out[:n]) == want
    print("pageable, runtime      : min %.3f median %.3f max %.3f mean %.3f ms  (%.0f MB/s)" % (lo, med, hi, mean, args.size / mean / 1e3))
    ctx.close()
    for t in [int(x) for x in args.threads.split(",")]:
        ctx = da.Context(0)
        ctx.reserve(args.size, host_api=True)
        ctx.config(da.Context.CFG_HOST_THREADS, t)
        n, lo, med, hi, mean = run(ctx, p_in.ctypes.data, p_out.ctypes.data)
        assert bytes(p_out[:n]) == want and ctx.info()["host_path"] == 7
        print("pageable, %2d threads   : min %.3f median %.3f max %.3f mean %.3f ms  (%.0f MB/s)" % (t, lo, med, hi, mean, args.size / mean / 1e3))
        ctx.close()


if __name__ == "__main__":
    main()
