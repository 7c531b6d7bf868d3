"""Development aid (GPU box): wall clock of the HOST-buffer call (mi355_deflate_encode: what deflate_bytes() binds) on small
inputs, pageable and page-locked buffers, beside the resident call.  usage: small_host_call.py"""
import ctypes as C, os, statistics, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "deflate-rs_amd")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import datagen, deflate_amd as da
ctx = da.Context(0)
cases = [("pg11 167 KB", open(os.path.join(ROOT, "tests/golden/ref_inputs/pg11.txt"), "rb").read()),
         ("text 32 KB", datagen.text_like(32768, 5)), ("text 2 MB", datagen.text_like(2_000_000, 2))]
for name, data in cases:
    n = len(data)
    cap = da.bound(n) + 16
    res = []
    for kind in ("pageable", "page-locked"):
        if kind == "pageable":
            src = (C.c_uint8 * n).from_buffer_copy(data)
            dst = (C.c_uint8 * cap)()
            ip, op = C.addressof(src), C.addressof(dst)
        else:
            tin = torch.frombuffer(bytearray(data), dtype=torch.uint8).pin_memory()
            tout = torch.empty(cap, dtype=torch.uint8).pin_memory()
            ip, op = tin.data_ptr(), tout.data_ptr()
        for _ in range(5):
            k = ctx.encode_host_ptr(ip, n, op, cap)
        ws = []
        for _ in range(50):
            t0 = time.perf_counter()
            k = ctx.encode_host_ptr(ip, n, op, cap)
            ws.append((time.perf_counter() - t0) * 1e3)
        res.append("%s %.3f ms (min %.3f)" % (kind, statistics.median(ws), min(ws)))
    t = torch.frombuffer(bytearray(data), dtype=torch.uint8).cuda()
    out = torch.empty(cap, dtype=torch.uint8, device="cuda")
    for _ in range(5):
        ctx.encode_device(t.data_ptr(), n, out.data_ptr(), cap)
    ws = []
    for _ in range(50):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        ctx.encode_device(t.data_ptr(), n, out.data_ptr(), cap)
        ws.append((time.perf_counter() - t0) * 1e3)
    print("%-12s -> %6d bytes   host call: %s   resident %.3f ms" % (name, k, "   ".join(res), statistics.median(ws)))
