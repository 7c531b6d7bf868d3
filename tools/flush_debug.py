"""Debug aid (GPU box): streams with flush points, HIP vs oracle, first differing token.
usage: flush_debug.py LEVEL CHUNK CUT [CUT ...]"""
import io, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "deflate-rs_amd")); sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import datagen, tokdump, deflate_amd as da, oracle_binding as ob
LV = {"fast": (1, 0, 0), "default": (128, 32, 1), "best": (1768, 128, 1), "rle": (0, 0, 1)}
level, chunk, cuts = sys.argv[1], int(sys.argv[2]), [int(x) for x in sys.argv[3:]]
c, l, m = LV[level]
ctx = da.Context(0)
texts = [datagen.text_like(260000, 41), datagen.mixed(200000, 12), datagen.rng_bytes(90000, 13), bytes(150000),
         (datagen.rng_bytes(300, 3) * 700)[:200000], (datagen.rng_bytes(4099, 4) * 40)[:150000]]
for ti, data in enumerate(texts):
    enc = da.DeflateEncoder(io.BytesIO(), da.CompressionOptions(c, l, m), ctx); ref = ob.Stream(ob.make_opts(c, l, m, 0))
    def put(piece):
        step = chunk or max(len(piece), 1); i = 0
        while i < len(piece):
            j = i + step
            if len(piece) - j == 1: j += 1
            enc.write_all(piece[i:j]); ref.write_all(piece[i:j]); i = j
    prev = 0
    for x in cuts:
        put(data[prev:x]); enc.flush(); ref.flush(); prev = x
    put(data[prev:])
    got = enc.finish().getvalue(); exp = ref.finish()
    if got == exp:
        print(ti, "OK", len(got)); continue
    print(ti, "DIFF", len(got), len(exp), "first token diff (index, hip, oracle):", tokdump.first_diff(got, exp))
    gb = tokdump.tokens(got); eb = tokdump.tokens(exp)
    print("  hip blocks   ", [(b["btype"], b["pos"], len(b["toks"])) for b in gb][:12])
    print("  oracle blocks", [(b["btype"], b["pos"], len(b["toks"])) for b in eb][:12])
