"""Development aid (run on the GPU box): a ladder of inputs through the HIP path, diffed against the
oracle with block-level localisation.  Not part of the product."""
import glob
import os
os.environ.setdefault("MI355_STAGE_CLOCKS", "1")  # (this aid reads the per-stage clocks: on for calls of every size)
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "deflate-rs_amd"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import datagen
import deflate_amd as da
import oracle_binding as ob

FIX = os.path.join(ROOT, "tests", "golden", "ref_inputs")
LV = {"fast": (1, 0, 0), "default": (128, 32, 1), "best": (1768, 128, 1), "rle": (0, 0, 1), "huff": (0, 0, 0)}
ctx = da.Context(0)
bad = 0


def check(name, data, levels=LV):
    global bad
    for ln, (c, l, m) in levels.items():
        ref = ob.encode(data, opts=ob.make_opts(c, l, m))
        rb = ob.trace_blocks()
        t0 = time.time()
        try:
            out = ctx.encode(data, da.CompressionOptions(c, l, m), compat=1)
        except Exception as e:
            print("%-28s %-8s EXC %s" % (name, ln, e))
            bad += 1
            continue
        dt = time.time() - t0
        ok = out == ref
        info = ctx.info()
        print("%-28s %-8s %s n=%d out=%d ref=%d blocks=%d T=%d q1=%d wall=%.1fms gpu=%.2fms match=%.2fms" % (
            name, ln, "OK " if ok else "BAD", len(data), len(out), len(ref), info["n_blocks"], info["n_tokens"],
            info["q1_rewarm"], dt * 1e3, info["total_ms"], info["match_ms"]))
        if not ok:
            bad += 1
            bl = ctx.blocks()
            for i, (a, b) in enumerate(zip(bl, rb)):
                if a != b:
                    print("   first differing block %d: gpu %s  ref %s" % (i, a, b))
                    break
            else:
                print("   block layouts equal (%d vs %d blocks); first differing byte:" % (len(bl), len(rb)),
                      next((i for i, (x, y) in enumerate(zip(out, ref)) if x != y), None))


check("empty", b"")
check("one", b"a")
check("short", bytes([10, 10, 10, 10, 10, 55]))
check("short.bin", open(os.path.join(FIX, "short.bin"), "rb").read())
check("zeros1000", bytes(1000))
check("zeros100000", bytes(100000))
check("pg11", open(os.path.join(FIX, "pg11.txt"), "rb").read())
check("issue18", open(os.path.join(FIX, "issue_18_201911.bin"), "rb").read())
check("dump.bin", open(os.path.join(FIX, "dump.bin"), "rb").read())
for f in sorted(glob.glob(os.path.join(FIX, "afl", "*")))[:6]:
    check(os.path.basename(f)[:18], open(f, "rb").read(), {k: LV[k] for k in ("default", "fast")})
check("rand100k", datagen.rng_bytes(100000, 1))
check("mixed300k", datagen.mixed(300000, 3))
check("text2M", datagen.text_like(2000000, 2), {k: LV[k] for k in ("default", "fast", "rle")})
check("text300k-best", datagen.text_like(300000, 4), {"best": LV["best"]})
print("FAILURES: %d" % bad)
sys.exit(1 if bad else 0)
