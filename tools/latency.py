"""Development aid (GPU box): wall-clock latency of one resident encode for small inputs."""
import os, sys, time, statistics
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "deflate-rs_amd")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import datagen, deflate_amd as da
ctx = da.Context(0)
cases = [("pg11 167 KB", open(os.path.join(ROOT, "tests/golden/ref_inputs/pg11.txt"), "rb").read()),
         ("text 2 MB", datagen.text_like(2_000_000, 2)), ("text 16 MB", datagen.text_like(16_000_000, 3)),
         ("random 1 MB (Q1)", datagen.rng_bytes(1_000_000, 4)), ("random 2 MB (Q1)", datagen.rng_bytes(2_000_000, 6))]
for name, data in cases:
    n = len(data)
    t = torch.frombuffer(bytearray(data), dtype=torch.uint8).cuda()
    cap = da.bound(n) + 8
    out = torch.empty(cap, dtype=torch.uint8, device="cuda")
    for lv in (da.Compression.Default, da.Compression.Fast):
        # the wall clock as a caller sees it (a call runs without the per-stage events unless asked: MI355_CFG_STAGE_CLOCKS), then
        # the same call with the events on for the stage table
        res = {}
        for clocks in (0, 1):
            ctx.config(da.Context.CFG_STAGE_CLOCKS, clocks)
            for _ in range(3):
                ctx.encode_device(t.data_ptr(), n, out.data_ptr(), cap, lv)
            ws = []
            for _ in range(40):
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                ctx.encode_device(t.data_ptr(), n, out.data_ptr(), cap, lv)
                ws.append((time.perf_counter() - t0) * 1e3)
            res[clocks] = (statistics.median(ws), min(ws), ctx.info())
        ctx.config(da.Context.CFG_STAGE_CLOCKS, 0)
        w, wmin, _ = res[0]
        wc, _, i = res[1]
        print("%-18s %-8s wall %.3f ms (min %.3f)  | with the stage clocks: wall %.3f, gpu events %.3f ms, stages %s" % (
            name, lv.name, w, wmin, wc, i["total_ms"], {k: round(v, 3) for k, v in i["stage_ms"].items()}))
