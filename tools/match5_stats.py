"""Debug aid (GPU box): iteration counts and clock shares of k_match5 (match_queue.inc) from an instrumented build
(-DMI355_MATCH_STATS -DMI355_MATCH_PATH_DEFAULT=6, deflate-rs_amd/variants/libstats5.so).  usage: match5_stats.py [bytes] [level]"""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.environ["MI355_DEFLATE_LIB"] = os.environ.get("MI355_STATS_LIB", os.path.join(ROOT, "deflate-rs_amd", "variants", "libstats5.so"))
sys.path.insert(0, os.path.join(ROOT, "deflate-rs_amd")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import datagen, deflate_amd as da
n = int(sys.argv[1]) if len(sys.argv) > 1 else 20_000_000
level = sys.argv[2] if len(sys.argv) > 2 else "default"
lv = {"default": da.Compression.Default, "best": da.Compression.Best}[level]
data = datagen.text_like(n, 0x656E)
ctx = da.Context(0); L = da.load(); out = (C.c_ulonglong * 16)()
ctx.encode(data, lv); L.mi355_debug_match_stats(out, 1)
ctx.encode(data, lv); L.mi355_debug_match_stats(out, 1)
s = list(out); per = 64.0 / n
print("positions", n, "match_ms", ctx.info()["match_ms"])
print("per 64 positions: set-ups %.2f (%.1f lanes)  walks %.2f (%.1f lanes)  compares %.2f (%.1f lanes)  idle polls %.1f" % (
    s[0] * per, s[5] / max(1, s[0]), s[1] * per, s[3] / max(1, s[1]), s[2] * per, s[4] / max(1, s[2]), s[6] * per))
t = {k: s[i] for k, i in (("setup", 8), ("compare", 9), ("choose", 11), ("walk", 12), ("idle", 13))}
tot = sum(t.values())
print("clock shares: " + "  ".join("%s %.3f" % (k, v / tot) for k, v in t.items()), " wave-cycles per 64 positions %.0f" % (tot * per))
print("cycles per iteration: set-up %.0f  walk %.0f  compare %.0f" % (s[8] / max(1, s[0]), s[12] / max(1, s[1]), s[9] / max(1, s[2])))
