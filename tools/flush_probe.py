"""GPU box: which (data, level, flush point) combinations of the pattern write(F) flush write(1) write(rest) differ from the oracle."""
import io, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "deflate-rs_amd")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa: F401
import datagen, deflate_amd as da, oracle_binding as ob
LV = {"fast": (1, 0, 0), "default": (128, 32, 1), "best": (1768, 128, 1)}
ctx = da.Context(0)
n = 140000
datas = {"per1": bytes([7]) * n, "per3": (datagen.rng_bytes(3, 1) * n)[:n], "per300": (datagen.rng_bytes(300, 2) * n)[:n],
         "per4099": (datagen.rng_bytes(4099, 3) * n)[:n], "text": datagen.text_like(n, 4)}
for dk, data in datas.items():
    for lv, (c, l, m) in LV.items():
        bad = []
        for F in (5000, 32768, 40000, 65535, 65536, 65537, 70000, 100000):
            for k in (1, 2):
                enc = da.DeflateEncoder(io.BytesIO(), da.CompressionOptions(c, l, m), ctx)
                ref = ob.Stream(ob.make_opts(c, l, m, 0))
                for e in (enc, ref):
                    e.write_all(data[:F]); e.flush(); e.write_all(data[F:F + k]); e.write_all(data[F + k:])
                if enc.finish().getvalue() != ref.finish():
                    bad.append((F, k))
        print(dk, lv, "differs at", bad)
