"""Debug aid (GPU box): counters and clock shares of k_match3 from an instrumented build (-DMI355_MATCH_STATS
-DMI355_MATCH_PATH_DEFAULT=4, deflate-rs_amd/variants/libstats3.so).  usage: match3_stats.py [bytes] [level] [text|silesia]"""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.environ["MI355_DEFLATE_LIB"] = os.environ.get("MI355_STATS_LIB", os.path.join(ROOT, "deflate-rs_amd", "variants", "libstats3.so"))
sys.path.insert(0, os.path.join(ROOT, "deflate-rs_amd")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import datagen, deflate_amd as da
n = int(sys.argv[1]) if len(sys.argv) > 1 else 20_000_000
level = sys.argv[2] if len(sys.argv) > 2 else "default"
lv = {"default": da.Compression.Default, "best": da.Compression.Best, "fast": da.Compression.Fast}[level]
data = datagen.silesia_like(scale=n / 212.1e6) if (len(sys.argv) > 3 and sys.argv[3] == "silesia") else datagen.text_like(n, 0x656E)
n = len(data)
ctx = da.Context(0)
L = da.load()
out = (C.c_ulonglong * 16)()
ctx.encode(data, lv)
L.mi355_debug_match_stats(out, 1)
ctx.encode(data, lv)
L.mi355_debug_match_stats(out, 1)
s = list(out)
nb = s[0]
print("positions", n, "match_ms", ctx.info()["match_ms"], "batches", nb)
print("step blocks/batch %.2f  walking lanes per block %.1f   services/batch %.2f  lanes settled per service %.1f" % (
    s[1] / nb, s[7] / max(1, s[1]), s[2] / nb, s[3] / max(1, s[2])))
t = {k: s[i] for k, i in (("setup", 8), ("service", 9), ("steps", 12), ("result", 13))}
tot = sum(t.values())
print("clock shares: " + "  ".join("%s %.3f" % (k, v / tot) for k, v in t.items()), " cycles/batch %.0f" % (tot / nb))
