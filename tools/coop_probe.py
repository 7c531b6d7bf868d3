"""Development aid: one small encode through MI355_MATCH_PATH=3 (k_match_coop) against the oracle."""
import os, sys
os.environ["MI355_MATCH_PATH"] = "3"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "deflate-rs_amd")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import deflate_amd as da, oracle_binding as ob, datagen
ctx = da.Context(0)
for name, data in (("tiny", b"abcabcabcabcabcabc" * 10), ("text40k", datagen.text_like(40000, 1)), ("text300k", datagen.text_like(300000, 2))):
    print(name, "...", flush=True)
    out = ctx.encode(data, da.Compression.Default)
    print(name, out == ob.encode(data, level=ob.DEFAULT), ctx.info()["match_ms"], flush=True)
