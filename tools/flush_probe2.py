"""GPU box: first differing token of write(F) flush write(1) write(rest) on period-3 data."""
import io, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "deflate-rs_amd")); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "tools"))
import torch  # noqa
import datagen, tokdump, deflate_amd as da, oracle_binding as ob
ctx = da.Context(0)
n = 140000
data = (datagen.rng_bytes(3, 1) * n)[:n]
c, l, m = (1, 0, 0)
for F in (int(x) for x in sys.argv[1:]):
    enc = da.DeflateEncoder(io.BytesIO(), da.CompressionOptions(c, l, m), ctx); ref = ob.Stream(ob.make_opts(c, l, m, 0))
    for e in (enc, ref):
        e.write_all(data[:F]); e.flush(); e.write_all(data[F:F + 1]); e.write_all(data[F + 1:])
    got = enc.finish().getvalue(); exp = ref.finish()
    print("F", F, "same" if got == exp else ("DIFF first token diff (index, hip, oracle): %s" % (tokdump.first_diff(got, exp),)))
    if got != exp:
        tg = [t for b in tokdump.tokens(got) for t in b["toks"]]
        te = [t for b in tokdump.tokens(exp) for t in b["toks"]]
        i = next(k for k, (x, y) in enumerate(zip(tg, te)) if x != y)
        print("   hip   ", tg[max(0, i - 2): i + 4])
        print("   oracle", te[max(0, i - 2): i + 4])
