"""GPU box: randomized test of the host-buffer call on pageable and page-locked buffers (deflate_bounce.inc, run_streamed, the long
path) against the device-resident call of the same context -- which the parity suite holds to the oracle -- so that sizes of
tens and hundreds of MB can be drawn.  Sizes around the 1 MiB / 4 MiB / 16 MiB thresholds and up to 300 MB, every level with a
hash, raw / zlib / gzip, 1..12 host threads, the threads on or off, two contexts at once from two Python threads.
usage: fuzz_pageable.py [cases] [first_seed]"""
import os, random, sys, threading
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "deflate-rs_amd")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch
import datagen, deflate_amd as da

POOL = {}


def pool(kind):
    if kind not in POOL:
        POOL[kind] = {"text": lambda: datagen.text_like(120_000_000, 0xF00D), "mixed": lambda: datagen.mixed(60_000_000, 0xF00E),
                      "rng": lambda: datagen.rng_bytes(40_000_000, 0xF00F), "zeros": lambda: bytes(64_000_000),
                      "silesia": lambda: datagen.silesia_like(scale=0.5)}[kind]()
    return POOL[kind]


def one(ctx, rnd, tag):
    kind = rnd.choice(["text", "text", "mixed", "rng", "zeros", "silesia"])
    src = pool(kind)
    n = rnd.choice([rnd.randrange(1, 1 << 20), rnd.randrange((4 << 20) - 4096, (4 << 20) + 4096), rnd.randrange((16 << 20) - 70000, (16 << 20) + 70000),
                    rnd.randrange(1 << 20, 40 << 20), rnd.randrange(min(40 << 20, len(src) - 1), min(len(src), 110 << 20))])
    n = min(n, len(src))
    off = rnd.randrange(0, len(src) - n + 1)
    data = src[off:off + n]
    lv = rnd.choice([da.Compression.Default, da.Compression.Default, da.Compression.Fast, da.Compression.Best])
    wrapper = rnd.choice([0, 0, 1, 2])
    ctx.config(da.Context.CFG_HOST_BOUNCE, rnd.choice([1, 1, 1, 0]))
    d_in = torch.frombuffer(bytearray(data), dtype=torch.uint8).cuda()
    cap = da.bound(n) + 64
    d_out = torch.empty(cap, dtype=torch.uint8, device="cuda")
    want_n = ctx.encode_device(d_in.data_ptr(), n, d_out.data_ptr(), cap, lv, wrapper=wrapper)  # (wrapper 2: the blank gzip header)
    want = d_out[:want_n].cpu().numpy()
    form = rnd.choice(["pageable", "pageable", "pinned", "page-in", "page-out"])
    p_in = np.frombuffer(data, dtype=np.uint8).copy() if form in ("pageable", "page-in") else torch.frombuffer(bytearray(data), dtype=torch.uint8).pin_memory()
    p_out = np.full(cap, 0xA5, dtype=np.uint8) if form in ("pageable", "page-out") else torch.full((cap,), 0xA5, dtype=torch.uint8).pin_memory()
    ip = p_in.ctypes.data if isinstance(p_in, np.ndarray) else p_in.data_ptr()
    op = p_out.ctypes.data if isinstance(p_out, np.ndarray) else p_out.data_ptr()
    got_n = ctx.encode_host_ptr(ip, n, op, cap, lv, wrapper=wrapper)
    got = p_out if isinstance(p_out, np.ndarray) else p_out.numpy()
    ok = got_n == want_n and np.array_equal(got[:got_n], want) and bool(np.all(got[got_n:] == 0xA5))
    if not ok:
        print("DIFFERENT", tag, kind, n, off, lv, wrapper, form, got_n, want_n, ctx.info()["host_path"], flush=True)
    return ok


def main():
    cases = int(sys.argv[1]) if len(sys.argv) > 1 else 60
    seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    bad = [0]
    done = [0]

    def work(t):
        rnd = random.Random(seed0 * 1000 + t)
        ctx = da.Context(0)
        ctx.config(da.Context.CFG_HOST_THREADS, rnd.choice([1, 2, 4, 8, 12]))
        try:
            for k in range(cases // 2):
                if not one(ctx, rnd, (t, k)):
                    bad[0] += 1
                done[0] += 1
        finally:
            ctx.close()
    for kind in ("text", "mixed", "rng", "zeros", "silesia"):
        pool(kind)
    ts = [threading.Thread(target=work, args=(t,)) for t in range(2)]
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    print("fuzz_pageable: %d cases, %d different, first_seed %d" % (done[0], bad[0], seed0))
    sys.exit(1 if bad[0] else 0)


main()
