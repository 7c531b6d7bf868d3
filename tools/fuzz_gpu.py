"""GPU box: randomized differential test, HIP path vs oracle, through the streaming API.
Every case draws a data generator, a size, compression options, a wrapper and a write/flush/reset
pattern from a seeded RNG and compares the bytes.  usage: fuzz_gpu.py [cases] [first_seed]"""
import io, os, random, sys, zlib
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "deflate-rs_amd")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa: F401 (before the library, see tests/test_gpu_parity.py)
import datagen, deflate_amd as da, oracle_binding as ob


def make_data(rnd):
    kind = rnd.choice(["text", "mixed", "rng", "zeros", "period", "lowent", "runs", "records"])
    n = rnd.choice([0, 1, 2, 3, 5, 100, 1000, 31744, 32768, 65536, 65794]) if rnd.random() < 0.15 else rnd.randrange(1, 400000)
    seed = rnd.randrange(1 << 30)
    if n == 0:
        return kind, b""
    if kind == "text":
        d = datagen.text_like(n, seed)
    elif kind == "mixed":
        d = datagen.mixed(n, seed)
    elif kind == "rng":
        d = datagen.rng_bytes(n, seed)
    elif kind == "zeros":
        d = bytes(n)
    elif kind == "records":  # rows of one length that differ in a counter and a few narrow fields (k_match3_swz's epochs), between text
        import numpy as np
        width = rnd.choice([24, 40, 48, 64, 96, 100, 128, 256, 512, 1000])
        r = np.random.default_rng(seed)
        rows = n // width + 1
        a = np.tile(r.integers(0, 256, size=width, dtype=np.uint8), (rows, 1))
        a[:, 4:8] = np.arange(rows, dtype=np.uint32).view(np.uint8).reshape(rows, 4)
        cols = r.choice(np.arange(8, width), size=max(1, width // 8), replace=False)
        a[:, cols] = r.integers(0, 16, size=(rows, len(cols)), dtype=np.uint8)
        d = a.reshape(-1)[:n].tobytes()
        if rnd.random() < 0.5:
            k = rnd.randrange(0, max(1, n // 2))
            d = d[:k] + datagen.text_like(min(60000, n), seed ^ 5)[: n - k if n - k < 60000 else 60000] + d[k:]
            d = d[:n]
    elif kind == "period":
        per = rnd.choice([1, 2, 3, 7, 300, 4099, 32768, 32769])
        d = (datagen.rng_bytes(per, seed) * (n // per + 1))[:n]
    elif kind == "lowent":
        r = random.Random(seed)
        d = bytes(r.choice(b"abcd") for _ in range(min(n, 120000)))
    else:
        r = random.Random(seed)
        out = bytearray()
        while len(out) < n:
            out += bytes([r.randrange(256)]) * r.choice([1, 2, 3, 4, 10, 258, 259, 1000])
        d = bytes(out[:n])
    return kind, d


def make_opts(rnd):
    if rnd.random() < 0.6:
        return rnd.choice([(1, 0, 0), (128, 32, 1), (1768, 128, 1), (0, 0, 1), (0, 0, 0)])
    return (rnd.choice([1, 2, 7, 32, 128, 500, 1768, 4000]), rnd.choice([3, 4, 8, 31, 32, 33, 64, 128, 258, 1000, 40000]),
            rnd.choice([0, 1]))


def one(seed, ctx):
    rnd = random.Random(seed)
    kind, data = make_data(rnd)
    c, l, m = make_opts(rnd)
    if c >= 1000 and len(data) > 150000:
        data = data[:150000]
    wrapper = rnd.choice([0, 0, 1, 2])
    n = len(data)
    # a write / flush / reset script
    cuts = sorted(rnd.sample(range(n + 1), min(n + 1, rnd.choice([0, 0, 1, 2, 5])))) if n else []
    script = []
    prev = 0
    for cpos in cuts:
        script.append(("w", prev, cpos))
        script.append((rnd.choice(["f", "f", "n", "r"]),))
        prev = cpos
    script.append(("w", prev, n))
    chunk = rnd.choice([0, 0, 0, 3, 1500, 40000, 70000])
    one_byte_calls = rnd.random() < 0.4  # 1-byte write calls (after a flush they skew the hash chains, lz77.rs:605-614)
    cls = (da.DeflateEncoder, da.ZlibEncoder, da.GzEncoder)[wrapper]
    enc = cls(io.BytesIO(), da.CompressionOptions(c, l, m), ctx)
    ref = ob.Stream(ob.make_opts(c, l, m, wrapper))
    if wrapper == 2:
        ref.gzip_header(da.BLANK_GZIP_HEADER)
    outs_e, outs_r = [], []
    try:
        for op in script:
            if op[0] == "w":
                piece = data[op[1]:op[2]]
                step = chunk or max(len(piece), 1)
                i = 0
                while i < len(piece):
                    j = i + step
                    if len(piece) - j == 1 and not one_byte_calls:
                        j += 1
                    enc.write_all(piece[i:j]); ref.write_all(piece[i:j]); i = j
            elif op[0] == "f":
                enc.flush(); ref.flush()
            elif op[0] == "r":
                outs_r.append(ref.reset())
                if wrapper == 2:
                    ref.gzip_header(da.BLANK_GZIP_HEADER)
                outs_e.append(enc.reset(io.BytesIO()).getvalue())
        outs_r.append(ref.finish())
        outs_e.append(enc.finish().getvalue())
    except da.DeflateError as e:
        if e.code == da.E_UNSUPPORTED:
            return "refused"   # documented: flush after 1-2 bytes, 1-byte write after a flush, lazy_if_less_than < 3
        raise
    except ob.RefPanic:
        return "ref-panic"
    if outs_e != outs_r:
        return "DIFF kind=%s n=%d opts=%s wrapper=%d chunk=%d script=%s" % (kind, n, (c, l, m), wrapper, chunk, script[:12])
    return "ok"


def main():
    cases = int(sys.argv[1]) if len(sys.argv) > 1 else 300
    first = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    ctx = da.Context(0)
    tally = {}
    bad = []
    for seed in range(first, first + cases):
        r = one(seed, ctx)
        k = r.split()[0]
        tally[k] = tally.get(k, 0) + 1
        if k == "DIFF":
            bad.append((seed, r))
            print("seed", seed, r, flush=True)
    print("fuzz:", tally, "first_seed", first)
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
