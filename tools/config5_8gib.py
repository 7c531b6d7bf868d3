"""GPU box: BASELINE config 5 at its stated size -- 8 GiB of the web-text input -- timed and checked against the oracle's digest
(tests/golden/config5_digest.json): (a) one GPU, the range walk of mi355_deflate_encode_device; (b) mi355_deflate_encode_multi_device
over the box's GPUs (--virtual N: N ranks on device 0, what a one-GPU box can run).  Prints one JSON line.
    python tools/config5_8gib.py [--virtual 8]"""
import argparse
import hashlib
import json
import multiprocessing as mp
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in ("deflate-rs_amd", "tests"):
    sys.path.insert(0, os.path.join(ROOT, p))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--virtual", type=int, default=0, help="ranks on device 0 (0: one rank per device of the box)")
    args = ap.parse_args()
    import torch
    import datagen
    import deflate_amd as da
    gold = json.load(open(os.path.join(ROOT, "tests", "golden", "config5_digest.json")))["digests"]
    N = gold["raw"]["in_len"]
    t0 = time.time()
    d_in = torch.empty(N + 64, dtype=torch.uint8, device="cuda:0")
    d_in[N:] = 0
    hin = hashlib.sha256()
    with mp.get_context("spawn").Pool(min(48, max(2, (os.cpu_count() or 4) - 2))) as pool:
        pending, nxt, done, n_seg = [], 0, 0, N // datagen.WEB_SEGMENT
        while done < N:
            while len(pending) < 4 and nxt < n_seg:
                pending.append(pool.map_async(datagen.webtext_segment_bytes, range(nxt, min(nxt + 64, n_seg)), chunksize=2))
                nxt = min(nxt + 64, n_seg)
            b = b"".join(pending.pop(0).get())
            hin.update(b)
            d_in[done:done + len(b)] = torch.frombuffer(bytearray(b), dtype=torch.uint8).cuda()
            done += len(b)
    res = {"workload": "config 5: webtext, %d bytes, Compression::Default" % N, "input_is_the_digests": hin.hexdigest() == gold["raw"]["in_sha256"],
           "generate_s": round(time.time() - t0, 1)}

    def digest_of(d_out, n):
        h = hashlib.sha256()
        for i in range(0, n, 256 << 20):
            h.update(bytes(d_out[i:min(n, i + (256 << 20))].cpu().numpy()))
        return [n, h.hexdigest()]
    cap = da.bound(N) + 64
    d_out = torch.empty(cap, dtype=torch.uint8, device="cuda:0")
    ctx = da.Context(0)
    for rep in range(2):
        torch.cuda.synchronize()
        t1 = time.time()
        n = ctx.encode_device(d_in.data_ptr(), N, d_out.data_ptr(), cap, da.Compression.Default)
        dt = time.time() - t1
    res["one_gpu_range_walk"] = {"ms": round(dt * 1e3, 1), "MB/s": round(N / dt / 1e6, 1), "ranges": ctx.info()["passes"],
                                 "same_as_oracle": digest_of(d_out, n) == [gold["raw"]["out_len"], gold["raw"]["out_sha256"]]}
    ctx.close()
    n_dev = da.load().mi355_device_count()
    devs = [0] * args.virtual if args.virtual else list(range(n_dev))
    if len(devs) > 1:
        m = da.MultiGpu(devs)
        W = m.layout(N, 0)["n_ranks"]
        lay = [m.layout(N, r) for r in range(W)]
        if args.virtual:
            ptrs = [d_in.data_ptr() + L["g_lo"] for L in lay]
            keep = []
        else:
            keep = [torch.cat([d_in[L["g_lo"]:L["g_hi"]].to("cuda:%d" % (r % len(devs))),
                               torch.zeros(64, dtype=torch.uint8, device="cuda:%d" % (r % len(devs)))]) for r, L in enumerate(lay)]
            ptrs = [k.data_ptr() for k in keep]
        for rep in range(2):
            d_out.fill_(0x33)
            for d in set(devs):
                torch.cuda.synchronize(d)
            t1 = time.time()
            n = m.encode_device(ptrs, N, d_out.data_ptr(), cap, da.Compression.Default)
            dt = time.time() - t1
        res["multi"] = {"devices": devs, "ranks": W, "ms": round(dt * 1e3, 1), "MB/s": round(N / dt / 1e6, 1), "trace_ms": m.trace(),
                        "same_as_oracle": digest_of(d_out, n) == [gold["raw"]["out_len"], gold["raw"]["out_sha256"]]}
        m.close()
    print(json.dumps(res))


if __name__ == "__main__":
    main()
