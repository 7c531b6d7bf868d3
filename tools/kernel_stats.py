"""Development aid (GPU box): counters and clock shares of k_match3 and k_sort from the instrumented build
(`make -C deflate-rs_amd stats` -> variants/libstats.so, -DMI355_MATCH_STATS).
usage: kernel_stats.py [match|sort] [bytes] [default|best|fast] [text|silesia|zeros]"""
import ctypes as C, os, subprocess, sys
os.environ.setdefault("MI355_STAGE_CLOCKS", "1")  # (this aid reads the per-stage clocks: on for calls of every size)
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = os.environ.get("MI355_STATS_LIB", os.path.join(ROOT, "deflate-rs_amd", "variants", "libstats.so"))
if not os.path.exists(LIB):
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "deflate-rs_amd"), "-s", "stats"])
os.environ["MI355_DEFLATE_LIB"] = LIB
sys.path.insert(0, os.path.join(ROOT, "deflate-rs_amd")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import datagen, deflate_amd as da
what = sys.argv[1] if len(sys.argv) > 1 else "match"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 20_000_000
level = sys.argv[3] if len(sys.argv) > 3 else "default"
kind = sys.argv[4] if len(sys.argv) > 4 else "text"
lv = {"default": da.Compression.Default, "best": da.Compression.Best, "fast": da.Compression.Fast}[level]
data = {"silesia": lambda: datagen.silesia_like(scale=n / 212.1e6), "zeros": lambda: bytes(n),
        "text": lambda: datagen.text_like(n, 0x656E)}[kind]()
n = len(data)
ctx = da.Context(0)
L = da.load()
out = (C.c_ulonglong * 16)()
read = L.mi355_debug_match_stats if what.startswith("match") else L.mi355_debug_sort_stats
for _ in range(2):  # (the second run is the one reported: the first one warms the context up)
    ctx.encode(data, lv)
    read(out, 1)
s = list(out)
if what == "match-counts":  # (the counters' build, run by the "match" call below)
    nb = s[0]
    print("step blocks/batch %.2f  walking lanes per block %.1f   services/batch %.2f  lanes settled per service %.1f" % (
        s[1] / nb, s[7] / max(1, s[1]), s[2] / nb, s[3] / max(1, s[2])))
elif what == "match":
    nb = s[0]
    print("positions", n, "match_ms", ctx.info()["match_ms"], "batches", nb)
    sys.stdout.flush()
    cnt = os.path.join(os.path.dirname(LIB), "libstats_cnt.so")
    if os.path.exists(cnt):
        subprocess.call([sys.executable, os.path.abspath(__file__), "match-counts"] + sys.argv[2:], env=dict(os.environ, MI355_STATS_LIB=cnt))
    t = {k: s[i] for k, i in (("setup", 8), ("service", 9), ("steps", 12), ("result", 13))}
    tot = sum(t.values())
    print("clock shares: " + "  ".join("%s %.3f" % (k, v / tot) for k, v in t.items()), " cycles/batch %.0f" % (tot / nb))
    print("(the instrumented kernel is not the product's: at the walk's limit of scalar registers its counters spill, and a clock read\n"
          " waits for the LDS reads in flight behind the step block -- that wait lands in `steps` here and in the service's first use in the\n"
          " product.  Until round 5's scalar-instruction work the same build read set-up 15 / service 51 / steps 34 %, 22 321 cycles a batch.)")
else:
    names = ["hash+hist", "bucket starts", "p1 count", "p1 offsets", "p1 scatter", "p2 count", "p2 offsets", "p2 scatter"]
    s = s[:8]
    tot = sum(s)
    print("links_ms", ctx.info()["stage_ms"]["links"], "cycles per epoch %.0f" % (tot / ((n + 32767) // 32768)))
    print("  ".join("%s %.3f" % (k, v / tot) for k, v in zip(names, s)))
