"""GPU box: differential test of write / flush patterns with tiny gaps (1-3 byte writes between sync flushes, at
the start of a stream, inside and beyond the first window, around the window edge) against the oracle.
usage: fuzz_flush_gaps.py [cases] [first_seed] [big]   (big: 400 KB inputs, flush points up to the seventh window)"""
import io, os, random, sys, zlib
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "deflate-rs_amd")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa: F401
import datagen, deflate_amd as da, oracle_binding as ob

LV = {"fast": (1, 0, 0), "default": (128, 32, 1), "best": (1768, 128, 1), "rle": (0, 0, 1)}


def main():
    cases = int(sys.argv[1]) if len(sys.argv) > 1 else 300
    first = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    big = len(sys.argv) > 3
    ctx = da.Context(0)
    res = {"ok": 0, "refused": 0}
    for seed in range(first, first + cases):
        rnd = random.Random(seed)
        kind = rnd.choice(["per", "text", "zeros", "rng"])
        n = 400000 if big else 140000
        data = {"per": (datagen.rng_bytes(rnd.choice([1, 3, 300, 4099]), seed) * n)[:n], "text": datagen.text_like(n, seed),
                "zeros": bytes(n), "rng": datagen.rng_bytes(n, seed)}[kind]
        pre = rnd.choice([0, 1, 2, 3, 4, 100, 5000, 30000, 32765, 32766, 32767, 32768, 32769, 32770, 40000, 65535, 65536, 65537, 70000])
        if big:
            pre = rnd.choice([98302, 98303, 98304, 98305, 98306, 120000, 131071, 131072, 131073, 163840, 200000, 229375, 229376, 229377, 65794, 66052])
        ops, pos = [], 0
        if pre:
            ops.append(pre); pos = pre
        for _ in range(rnd.randrange(1, 9)):
            r = rnd.random()
            if r < 0.45:
                ops.append("F")
            else:
                k = rnd.choice([1, 1, 1, 2, 2, 3, 4, rnd.randrange(1, 2000)])
                ops.append(k); pos += k
        tail = rnd.choice([0, 1, 2, 50, 3000, 70000])
        if tail:
            ops.append(min(tail, n - pos))
        lv = rnd.choice(list(LV))
        c, l, m = LV[lv]
        wrapper = rnd.choice([0, 1])
        enc = (da.ZlibEncoder if wrapper else da.DeflateEncoder)(io.BytesIO(), da.CompressionOptions(c, l, m), ctx)
        ref = ob.Stream(ob.make_opts(c, l, m, wrapper))
        p = 0
        try:
            for op in ops:
                if op == "F":
                    enc.flush(); ref.flush()
                else:
                    enc.write_all(data[p:p + op]); ref.write_all(data[p:p + op]); p += op
            got = enc.finish().getvalue()
        except da.DeflateError as e:
            assert e.code == da.E_UNSUPPORTED, e
            enc._done = True
            res["refused"] += 1
            continue
        want = ref.finish()
        if got != want:
            print("DIFFERENT seed", seed, kind, lv, "wrapper", wrapper, ops, len(got), len(want))
            sys.exit(1)
        res["ok"] += 1
    print("fuzz_flush_gaps:", res, "first_seed", first)


main()
