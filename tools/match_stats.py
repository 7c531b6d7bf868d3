"""Debug aid (GPU box): per-launch counters of k_match from the instrumented build
(deflate-rs_amd/variants/libstats.so, -DMI355_MATCH_STATS).  usage: match_stats.py [bytes] [level]"""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.environ["MI355_DEFLATE_LIB"] = os.environ.get("MI355_STATS_LIB", os.path.join(ROOT, "deflate-rs_amd", "variants", "libstats.so"))
sys.path.insert(0, os.path.join(ROOT, "deflate-rs_amd")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import datagen, deflate_amd as da
n = int(sys.argv[1]) if len(sys.argv) > 1 else 20_000_000
level = sys.argv[2] if len(sys.argv) > 2 else "default"
lv = {"default": da.Compression.Default, "best": da.Compression.Best, "fast": da.Compression.Fast}[level]
data = datagen.text_like(n, 0x656E)
ctx = da.Context(0)
L = da.load()
out = (C.c_ulonglong * 16)()
ctx.encode(data, lv)
L.mi355_debug_match_stats(out, 1)
ctx.encode(data, lv)
L.mi355_debug_match_stats(out, 1)
s = list(out)
U = 2
pos = n
print("positions", pos, "match_ms", ctx.info()["match_ms"])
print("lane-steps/pos %.1f   slot-steps/pos %.1f" % (s[0] / pos, U * s[0] / pos))
tot = s[1] + s[4] + s[5] + s[6]
print("slot states: WALK %.3f PARK %.3f FIN %.3f IDLE %.3f  (walk visits/pos %.1f)" % (s[1] / tot, s[4] / tot, s[5] / tot, s[6] / tot, s[1] / pos))
print("services per lane-step %.3f, ext rounds per service %.2f" % (s[2] / s[0], s[3] / max(s[2], 1)))
wgs = s[9]
print("workgroups %d: mean lane steps %.0f, mean of per-WG max %.0f  -> tail factor %.2f" % (wgs, s[0] / (wgs * 1024), s[8] / wgs, (s[8] / wgs) / (s[0] / (wgs * 1024))))
nw = wgs * 16
print("wall clock per WG (10 ns ticks): last wave %.0f, mean wave %.0f -> tail factor %.3f; staging %.0f" % (s[10] / wgs, s[11] / nw, (s[10] / wgs) / (s[11] / nw), s[12] / wgs))
