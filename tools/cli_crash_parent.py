"""development aid (GPU box): the C example run the way tests/test_gpu_parity.py::test_c_example_program runs it --
as a child of a Python process that holds a context of the library on the same GPU -- N times per configuration,
under the fault trap of tools/probes/segv_trap.c.  usage: cli_crash_parent.py [N = 200]"""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "deflate-rs_amd")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import deflate_amd as da
N = int(sys.argv[1]) if len(sys.argv) > 1 else 200
subprocess.run(["gcc", "-g", "-O2", "-std=c99", "-I", os.path.join(ROOT, "include"), os.path.join(ROOT, "examples", "mi355_deflate_cli.c"),
                "-L", os.path.join(ROOT, "deflate-rs_amd"), "-lmi355deflate", "-Wl,-rpath," + os.path.join(ROOT, "deflate-rs_amd"),
                "-o", "/tmp/cli"], check=True)
subprocess.run(["gcc", "-O1", "-g", "-shared", "-fPIC", "-o", "/tmp/segv_trap.so", os.path.join(ROOT, "tools", "probes", "segv_trap.c"), "-ldl"], check=True)
src = os.path.join(ROOT, "tests", "golden", "ref_inputs", "pg11.txt")
data = open(src, "rb").read()
ctx = da.Context(0)
ctx.encode(data, da.Compression.Default)
env = dict(os.environ, LD_PRELOAD="/tmp/segv_trap.so")
fails = 0
runs = 0
for i in range(N):
    for flag, lvl in (("-raw", "-default"), ("-zlib", "-best"), ("-gzip", "-fast")):
        for extra in ([], ["-chunk", "5000"]):
            if i % 8 == 0:
                ctx.encode(data, da.Compression.Best)  # the parent keeps using the GPU between children
            r = subprocess.run(["/tmp/cli", flag, lvl] + extra + [src, "/tmp/out.bin"], capture_output=True, env=env)
            runs += 1
            if r.returncode != 0:
                fails += 1
                print("=== run %d %s %s %s rc=%d\n%s" % (i, flag, lvl, extra, r.returncode, r.stderr.decode(errors="replace")[-6000:]), flush=True)
print("cli under a parent context: fails=%d of %d" % (fails, runs))
