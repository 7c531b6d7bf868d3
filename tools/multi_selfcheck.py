"""One line for a box with several GPUs: does mi355_deflate_encode_multi -- ONE input over all devices in one call -- produce
the reference's stream there?  Every device of the node (or --gpus N of them) takes a range of each input; the stream is
compared with the CPU oracle's (oracle/, the checker), raw / zlib / gzip, host buffers and resident shards.  This is the run
that exercises what one-GPU boxes cannot: hipDeviceEnablePeerAccess, hipMemcpyPeerAsync between distinct devices, the seam
words on rank 0's device.
    python tools/multi_selfcheck.py [--gpus N] [--mb-per-gpu 32]
Prints one JSON line; exit code 0 = every stream identical."""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in ("deflate-rs_amd", "tests"):
    sys.path.insert(0, os.path.join(ROOT, p))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=0, help="devices to use (0: all)")
    ap.add_argument("--mb-per-gpu", type=int, default=32)
    ap.add_argument("--no-rccl", action="store_true", help="skip the resident case with the RCCL stitch (MI355_CFG_MULTI_STITCH = 1)")
    args = ap.parse_args()
    import torch  # noqa: F401  (before the library: tests/test_gpu_parity.py)
    import datagen
    import deflate_amd as da
    import oracle_binding as ob

    n_dev = da.load().mi355_device_count()
    n = args.gpus or n_dev
    if n < 1 or n > n_dev:
        print(json.dumps({"ok": False, "error": "%d devices asked for, %d present" % (n, n_dev)}))
        return 2
    m = da.MultiGpu(list(range(n)))
    per = args.mb_per_gpu << 20
    inputs = [("text", datagen.text_like(per * n + 12345, 0x51)),
              ("noise", datagen.rng_bytes(max(per // 4, 1 << 20) * n + 77, 0x52)),
              ("zeros", bytes(per * n)),
              ("mixed", datagen.mixed(max(per // 2, 1 << 20) * n, 0x53))]
    c, l, mt = 128, 32, 1
    res = {"ok": True, "devices": n, "cases": []}
    t_all = time.time()
    for name, data in inputs:
        for wrapper in (0, 1, 2):
            if wrapper and name not in ("text", "mixed"):
                continue
            want = ob.encode(data, opts=ob.make_opts(c, l, mt, wrapper)) if wrapper < 2 else ob.encode_gzip(
                data, da.BLANK_GZIP_HEADER, opts=ob.make_opts(c, l, mt, 0))
            t0 = time.time()
            got = m.encode(data, da.Compression.Default, wrapper=wrapper)
            dt = time.time() - t0
            same = got == want
            res["cases"].append({"input": name, "bytes": len(data), "wrapper": wrapper, "form": "host", "same": same,
                                 "ranks": m.layout(len(data), 0)["n_ranks"], "ms": round(dt * 1e3, 2)})
            res["ok"] = res["ok"] and same
        # resident shards: every rank's bytes on its own device, the stream assembled on rank 0's by peer copies
        W = m.layout(len(data), 0)["n_ranks"]
        lay = [m.layout(len(data), r) for r in range(W)]
        bufs = []
        for r, L in enumerate(lay):
            dev = "cuda:%d" % (r % n)
            bufs.append(torch.frombuffer(bytearray(data[L["g_lo"]:L["g_hi"]]) + bytearray(64), dtype=torch.uint8).to(dev))
        cap = da.bound(len(data)) + 64
        d_out = torch.full((cap,), 0xAA, dtype=torch.uint8, device="cuda:0")
        for d in range(n):
            torch.cuda.synchronize(d)
        want = ob.encode(data, opts=ob.make_opts(c, l, mt, 0))
        for stitch in ((0,) if args.no_rccl else (0, 1)):  # peer copies, then ncclSend / ncclRecv
            m.config(da.Context.CFG_MULTI_STITCH, stitch)
            d_out.fill_(0xAA)
            for d in range(n):
                torch.cuda.synchronize(d)
            t0 = time.time()
            k = m.encode_device([b.data_ptr() for b in bufs], len(data), d_out.data_ptr(), cap, da.Compression.Default)
            dt = time.time() - t0
            same = bytes(d_out[:k].cpu().numpy()) == want
            res["cases"].append({"input": name, "bytes": len(data), "wrapper": 0, "form": "resident, " + ("RCCL stitch" if stitch else "peer copies"),
                                 "same": same, "ranks": W, "ms": round(dt * 1e3, 2), "MB/s": round(len(data) / dt / 1e6, 1), "trace": m.trace()})
            res["ok"] = res["ok"] and same
        m.config(da.Context.CFG_MULTI_STITCH, 0)
    res["seconds"] = round(time.time() - t_all, 1)
    m.close()
    print(json.dumps(res))
    return 0 if res["ok"] else 1


if __name__ == "__main__":
    sys.exit(main())
