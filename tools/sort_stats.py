"""Debug aid (GPU box): clock shares of the phases of k_sort (instrumented build, MI355_MATCH_PATH must select the sorted path)."""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.environ["MI355_DEFLATE_LIB"] = os.environ.get("MI355_STATS_LIB", os.path.join(ROOT, "deflate-rs_amd", "variants", "libstats.so"))
sys.path.insert(0, os.path.join(ROOT, "deflate-rs_amd")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import datagen, deflate_amd as da
n = 20_000_000
data = bytes(n) if len(sys.argv) > 1 and sys.argv[1] == "zeros" else datagen.text_like(n, 0x656E)
ctx = da.Context(0); L = da.load(); out = (C.c_ulonglong * 16)()
ctx.encode(data, da.Compression.Default); L.mi355_debug_sort_stats(out, 1)
ctx.encode(data, da.Compression.Default); L.mi355_debug_sort_stats(out, 1)
s = list(out)[:8]
names = ["hash+hist", "bucket starts", "p1 count | registers", "p1 offsets | turn", "p1 scatter | scatter", "p2 count", "p2 offsets", "p2 scatter"]  # (two-pass modes | k_sort<2>)
tot = sum(s); ne = (n + 32767) // 32768
print("links_ms", ctx.info()["stage_ms"], "cycles per epoch %.0f" % (tot / ne))
print("  ".join("%s %.3f" % (k, v / tot) for k, v in zip(names, s)))
