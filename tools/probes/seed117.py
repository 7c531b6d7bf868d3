"""development aid (GPU box): seed 117 of tools/fuzz_gpu.py step by step"""
import io, os, random, sys, zlib
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
for p in ("deflate-rs_amd", "tests", "tools"):
    sys.path.insert(0, os.path.join(ROOT, p))
import torch  # noqa
import deflate_amd as da, oracle_binding as ob, fuzz_gpu, tokdump
rnd = random.Random(117)
kind, data = fuzz_gpu.make_data(rnd)
c, l, m = fuzz_gpu.make_opts(rnd)
print(kind, len(data), c, l, m)
ctx = da.Context(0)
cuts = [85959, 106241, 109989]
for wrapper in (0,):
    enc = da.DeflateEncoder(io.BytesIO(), da.CompressionOptions(c, l, m), ctx)
    ref = ob.Stream(ob.make_opts(c, l, m, wrapper))
    prev = 0
    for k, cut in enumerate(cuts):
        enc.write(data[prev:cut]); ref.write_all(data[prev:cut])
        if k < 2:
            enc.flush(); ref.flush()
        else:
            enc.finish(); ref.finish()
        a, b = enc._w.getvalue(), ref.output()
        print("after", cut, "same:", a == b, len(a), len(b), {k2: v for k2, v in ctx.info().items() if k2 in ("q1_rewarm", "passes", "n_blocks", "n_tokens")})
        if a != b:
            d = next(i for i in range(min(len(a), len(b))) if a[i] != b[i])
            print("first differing byte", d)
            # the tokens of both (sync-flushed streams decode up to the marker)
            try:
                ta = [t for bl in tokdump.tokens(a + b"\x01\x00\x00\xff\xff") for t in bl["toks"]]
                tb = [t for bl in tokdump.tokens(b + b"\x01\x00\x00\xff\xff") for t in bl["toks"]]
                i = next(i for i in range(min(len(ta), len(tb))) if ta[i] != tb[i])
                print("first differing token", i, "gpu", ta[i - 2:i + 3], "oracle", tb[i - 2:i + 3])
            except Exception as e:
                print("decode failed", e)
            break
        prev = cut
# one shot
got = ctx.encode(data, da.CompressionOptions(c, l, m))
want = ob.encode(data, opts=ob.make_opts(c, l, m))
print("one shot same:", got == want, ctx.info()["q1_rewarm"])
got = ctx.encode(data[:85959], da.CompressionOptions(c, l, m))
want = ob.encode(data[:85959], opts=ob.make_opts(c, l, m))
print("one shot of the first part same:", got == want, ctx.info()["q1_rewarm"])
