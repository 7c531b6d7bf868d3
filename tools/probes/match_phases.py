"""Development aid (GPU box): where a wave of k_match3 spends its clocks on a SMALL input (the instrumented build,
-DMI355_MATCH_STATS=1): staging + barrier, set-up of its batches, services, step blocks, results -- per wave, in microseconds
at the clock the counter runs at (printed).  usage: match_phases.py [file | text:<bytes>]"""
import ctypes as C, os, subprocess, sys, time
os.environ.setdefault("MI355_STAGE_CLOCKS", "1")
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
LIB = os.path.join(ROOT, "deflate-rs_amd", "variants", "libstats.so")
os.environ["MI355_DEFLATE_LIB"] = LIB
sys.path.insert(0, os.path.join(ROOT, "deflate-rs_amd")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import datagen, deflate_amd as da
arg = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "tests/golden/ref_inputs/pg11.txt")
data = datagen.text_like(int(arg[5:]), 3) if arg.startswith("text:") else open(arg, "rb").read()
n = len(data)
ctx = da.Context(0)
L = da.load()
out = (C.c_ulonglong * 16)()
t = torch.frombuffer(bytearray(data), dtype=torch.uint8).cuda()
cap = da.bound(n) + 8
o = torch.empty(cap, dtype=torch.uint8, device="cuda")
for _ in range(3):
    ctx.encode_device(t.data_ptr(), n, o.data_ptr(), cap, da.Compression.Default)
    L.mi355_debug_match_stats(out, 1)
s = list(out)
waves = max(1, s[4])
i = ctx.info()
names = [("stage + barrier", 10), ("set-up", 8), ("service", 9), ("steps", 12), ("result", 13), ("tail", 11), ("turn round", 14)]
tot = sum(s[k] for _, k in names)
print("%d bytes, match stage %.1f us, %d waves, %d batches; per wave (cycles of the counter):" % (n, i["stage_ms"]["match"] * 1e3, waves, s[0]))
for nm, k in names:
    print("  %-16s %9.0f  %5.1f %%" % (nm, s[k] / waves, 100.0 * s[k] / max(1, tot)))
print("  %-16s %9.0f" % ("sum", tot / waves))
