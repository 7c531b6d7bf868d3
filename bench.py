"""bench.py -- MB/s of raw input encoded at Compression::Default on MI355X, next to the CPU oracle.

One step = one pass of the whole encode path (links -> match -> parse -> blocks -> pack, plus the
stitch on N > 1) over one batch of synthetic input that is already resident in HBM.  Contract: see
the task statement; prints ONE JSON line on rank 0.

Workloads (BASELINE.json configs): "enwik8" = 100 000 000 bytes of enwik8-like text, Default,
dynamic-Huffman blocks (the configuration the metric is quoted on; default); "zeros" = 256 MiB zero
fill through the RLE path (config 2); "random" = 64 MiB noise (stored blocks); "silesia" = the
Silesia-like mix at Compression::Best (config 4); "webtext" = config 5: every rank owns 1 GiB of ONE
N GiB web-text input (generated per 1 MiB segment, so a rank makes exactly its own part).

N > 1 (`python bench.py --gpus N`, with or without torch.distributed.run around it: without, bench.py launches itself under it):
one process per GPU over RCCL, the workload defaults to config 5 -- every rank owns 1 GiB of ONE N GiB web-text input, 8 GiB at
N = 8 -- and the line says what RCCL saw (rccl_ranks), how the stream was stitched, rank 0's phases, and whether the stitched
stream is the oracle's (committed digests, tests/golden/config5_digest*.json).  --virtual: the N ranks share the GPUs the box has
(gloo carries the exchanges) -- the dry run of a one-GPU box.  --single-process: the same sharding inside ONE call of the C ABI.

Besides the contract's fields the line carries: value_host_api (the drop-in call on pinned host buffers,
H2D and D2H inside the timed region), value_host_api_pageable (the same call on the memory a drop-in caller has: a plain
bytearray in and out -- the context's host threads carry it; beside it the runtime's own copies, MI355_CFG_HOST_BOUNCE = 0), roofline.input_load (achieved HBM GB/s of the kernel that reads
the input coalesced), roofline.lds_bank_conflict_rate of the match compare and roofline.valu_issue (wave
instructions per input byte and the share of the SIMDs' cycles they take: what the dominant kernel is bound
by; from counters taken in the run, else from the committed PMC file),
value_as_called (the timed steps once more as a caller's context runs them: without the per-stage clocks this bench asks for -- five
events, 29 us of idle queue, in every timed step), small_call (the reference's 167 KB fixture pg11.txt: wall clock of one resident call and of one host-buffer call on pageable memory,
against the oracle), cpu_baseline as the median of five runs with its all-cores and zlib companions.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, "deflate-rs_amd"))
sys.path.insert(0, os.path.join(ROOT, "tests"))

HBM_PEAK_GBS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec


ENWIK8_PATHS = [os.environ.get("MI355_ENWIK8", ""), os.path.join(ROOT, "data", "enwik8"), "/data/enwik8"]
INPUT_NOTE = {}


def make_input(workload, size, rank):
    import datagen
    if workload == "enwik8":
        # the real enwik8 when the box has it (there is no network to fetch it: data/enwik8 or
        # $MI355_ENWIK8), else enwik8-like synthetic text of the same size; the bench line says which
        for p in ENWIK8_PATHS:
            if p and os.path.isfile(p) and os.path.getsize(p) >= size * (rank + 1):
                with open(p, "rb") as f:
                    f.seek(size * rank)
                    INPUT_NOTE["enwik8"] = "the real enwik8 (%s)" % p
                    return f.read(size)
        INPUT_NOTE["enwik8"] = "enwik8-like synthetic text (tests/datagen.py text_like; no data/enwik8 on this box)"
        return datagen.text_like(size, 0x656E77696B38 ^ rank)
    if workload == "zeros":
        return bytes(size)
    if workload == "random":
        return datagen.rng_bytes(size, 0x5EED0001 ^ rank)
    if workload == "silesia":  # BASELINE config 4: twelve pieces by entropy class, Silesia's file sizes
        d = datagen.silesia_like(0x53494C45 ^ rank)
        return d if size >= len(d) else d[:size]
    if workload == "webtext":  # BASELINE config 5: this rank's part of the one big input
        seg = datagen.WEB_SEGMENT
        if size >= (64 << 20) and size % seg == 0 and "torch" not in sys.modules:
            # (a GiB a rank: generated a MiB segment at a time on a few cores -- before torch and HIP are loaded, so forking is safe)
            import multiprocessing as mp
            first = rank * size // seg
            nproc = max(1, min(16, (os.cpu_count() or 2) // max(1, int(os.environ.get("LOCAL_WORLD_SIZE", "1")))))
            with mp.get_context("fork").Pool(nproc) as pool:
                return b"".join(pool.map(datagen.webtext_segment_bytes, range(first, first + size // seg), chunksize=8))
        return datagen.webtext(size, start=rank * size)
    raise SystemExit("unknown workload " + workload)


def cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def cpu_baselines(data, olvl, level_name):
    """The oracle (C++ restatement of deflate-rs, same algorithm) on the host cores, on a bounded sample:
    median of five single-thread runs; the same sample cut into one chunk per core and encoded by all cores
    at once (chunk-exact form, P2: what N independent reference encoders would do); system zlib -6 as an
    anchor that is NOT the reference."""
    import statistics
    import threading
    import zlib
    import oracle_binding as ob
    ncpu = os.cpu_count() or 1
    sample = data[: min(len(data), 16_000_000)]
    times = []
    for _ in range(5):
        t = time.perf_counter()
        ob.encode(sample, level=olvl)
        times.append(time.perf_counter() - t)
    one = len(sample) / statistics.median(times) / 1e6
    big = data[: min(len(data), ncpu * 8_000_000)]
    step = (len(big) + ncpu - 1) // ncpu
    chunks = [big[i:i + step] for i in range(0, len(big), step)]
    th = [threading.Thread(target=ob.encode, args=(c,), kwargs={"level": olvl}) for c in chunks]  # (ctypes drops the GIL)
    t = time.perf_counter()
    for x in th:
        x.start()
    for x in th:
        x.join()
    allc = len(big) / (time.perf_counter() - t) / 1e6
    t = time.perf_counter()
    zlib.compress(sample, 6)
    z6 = len(sample) / (time.perf_counter() - t) / 1e6
    return {"value": round(one, 2), "unit": "MB/s", "cores": 1, "kind": "port",
            "sample": "first %d bytes of the workload, %s, median of 5 runs of the single-threaded C++ restatement "
                      "of deflate-rs (oracle/)" % (len(sample), level_name),
            "runs_s": [round(x, 3) for x in times], "host_cores": ncpu, "cpu_model": cpu_model(),
            "all_cores": {"value": round(allc, 2), "unit": "MB/s", "cores": len(chunks),
                          "sample": "first %d bytes in %d chunks, one oracle encoder per core (chunk-exact, P2)" % (
                              len(big), len(chunks))},
            "zlib6_anchor": {"value": round(z6, 2), "unit": "MB/s", "cores": 1, "note": "system zlib -6, not the reference"}}


def single_process(args):
    """N GPUs of the node, ONE process: the library shards the input itself (row h of SURVEY section 8).  Weak scaling like
    the multi-process form: every device owns `size` bytes of one N x size input, resident in its HBM with its history and
    look-ahead; the stream lands in device 0's memory.  value = N x size bytes / step."""
    N = args.gpus
    size = args.size or {"enwik8": 100_000_000, "webtext": 1 << 30}.get(args.workload, 100_000_000)
    size = (size + 32767) // 32768 * 32768
    total = N * size
    part = {r: make_input(args.workload, size, r) for r in range(N)}  # (before torch and HIP are loaded: made on a pool of processes)
    import torch
    import deflate_amd as da
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the product has no CPU path")
    ndev = torch.cuda.device_count()
    if not args.virtual and ndev < N:
        raise SystemExit("--gpus %d but %d devices visible (use --virtual for a dry run on the devices there are)" % (N, ndev))
    devs = [r % ndev for r in range(N)]
    lvl = args.level or "default"
    options = {"default": da.CompressionOptions.default, "best": da.CompressionOptions.high,
               "fast": da.CompressionOptions.fast}[lvl]()
    level_name = {"default": "Compression::Default", "best": "Compression::Best", "fast": "Compression::Fast"}[lvl]
    m = da.MultiGpu(devs)
    stitch = args.stitch
    if stitch == "rccl":
        m.config(da.Context.CFG_MULTI_STITCH, 1)
    lay = [m.layout(total, r) for r in range(N)]
    assert lay[0]["n_ranks"] == N
    # every rank holds its own bytes [g_lo, g_hi): its part of the one input, the last 32 KiB of the part before, the first
    # 128 KiB of the part behind
    bufs = []

    def piece(r):
        return part[r]
    for r in range(N):
        L = lay[r]
        b = bytearray()
        p = L["g_lo"]
        while p < L["g_hi"]:
            q = p // size
            e = min(L["g_hi"], (q + 1) * size)
            b += piece(q)[p - q * size:e - q * size]
            p = e
        bufs.append(torch.frombuffer(b + bytearray(64), dtype=torch.uint8).to("cuda:%d" % devs[r]))
        for k in [k for k in part if k < r]:
            del part[k]
    part.clear()
    cap = da.bound(total) + 64
    d_out = torch.empty(cap, dtype=torch.uint8, device="cuda:%d" % devs[0])
    ptrs = [b.data_ptr() for b in bufs]
    n = 0
    try:
        n = m.encode_device(ptrs, total, d_out.data_ptr(), cap, options)
    except da.DeflateError as e:
        if stitch != "rccl":
            raise
        sys.stderr.write("bench.py: the RCCL stitch is not available here (%s): peer copies\n" % e)
        stitch = "peer"
        m.config(da.Context.CFG_MULTI_STITCH, 0)
    for _ in range(args.warmup):
        n = m.encode_device(ptrs, total, d_out.data_ptr(), cap, options)
    for d in set(devs):
        torch.cuda.synchronize(d)
    traces = []
    mm = []
    t0 = time.perf_counter()
    for _ in range(args.steps):
        n = m.encode_device(ptrs, total, d_out.data_ptr(), cap, options)
        traces.append(m.trace())
        mm.append(max(m.rank_info(r)["match_ms"] / max(1, m.rank_info(r)["match_launches"]) for r in range(N)))
    for d in set(devs):
        torch.cuda.synchronize(d)
    elapsed = time.perf_counter() - t0
    tr = {k: round(sum(t[k] for t in traces) / len(traces), 4) for k in traces[0]}
    algo = size + n // N
    k_ms = sum(mm) / len(mm)
    res = {"metric": "MB/s raw input encoded (%s) + compressed size vs ref" % level_name,
           "value": round(total * args.steps / elapsed / 1e6, 2), "unit": "MB/s", "n_gpus": N, "steps": args.steps,
           "warmup": args.warmup, "ms_per_step": round(elapsed * 1e3 / args.steps, 3), "higher_is_better": True,
           "scaling": "weak", "vs_baseline": None, "dtype": "u8",
           "data": "real" if "real" in INPUT_NOTE.get(args.workload, "") else "synthetic",
           "config": {"workload": "%s%s: %d bytes per GPU, %s, one %d-byte input sharded over %d %s in ONE process "
                                  "(mi355_deflate_encode_multi_device), stream-exact (P1), stitch by %s" % (
                                      args.workload, " = " + INPUT_NOTE[args.workload] if args.workload in INPUT_NOTE else "", size,
                                      level_name, total, N, "ranks that SHARE %d device(s) (dry run)" % ndev if ndev < N else "GPUs",
                                      "ncclSend / ncclRecv" if stitch == "rccl" else "peer copies"),
                      "bytes_per_gpu": size, "level": level_name, "parallelism": "shard%d" % N},
           "out_bytes": n, "ratio": round(n / total, 5),
           "roofline": {"bound": "hbm", "kernel": "k_match3", "achieved": round(algo / (k_ms * 1e-3) / 1e9, 2), "peak": HBM_PEAK_GBS,
                        "unit": "GB/s", "frac": round(algo / (k_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 5), "traffic": None,
                        "kernel_ms": round(k_ms, 3), "algorithmic_bytes_per_launch": algo},
           "multi_phases_ms_rank0": tr}
    si = m.stitch_info()
    res["stitch"] = si["stitch"]
    res["rccl_ranks"] = si["rccl_ranks"]  # (what the communicator of rank 0's device reports: one RCCL rank per distinct device)
    res["devices"] = ndev

    def parts():
        for i in range(0, n, 256 << 20):
            yield bytes(d_out[i:min(n, i + (256 << 20))].cpu().numpy())
    res.update(stream_check(args.workload, total, size, N, lvl, parts, n))
    m.close()
    print(json.dumps(res))
    quiet_stdout()


def quiet_stdout():
    """The JSON line is the last thing on stdout: librccl prints its version banner there when the process exits (every rank's
    lands on the launcher's stdout) -- from here on file descriptor 1 is /dev/null."""
    sys.stdout.flush()
    try:
        os.dup2(os.open(os.devnull, os.O_WRONLY), 1)
    except OSError:
        pass


def device_count():
    """HIP devices of the box, without loading torch (mi355_device_count of the C ABI; 0: no GPU or no HIP runtime)"""
    import deflate_amd as da
    try:
        return int(da.load().mi355_device_count())
    except Exception:
        return 0


def launch_plan(args, argv, n_dev):
    """`python bench.py --gpus N` without torch.distributed.run around it: the command that starts the N ranks -- what the
    driver runs for N > 1 -- and its environment.  A box with fewer GPUs than ranks runs it only as a dry run (--virtual):
    the ranks share the devices and the exchanges travel over gloo."""
    import socket
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        port = so.getsockname()[1]
    env = {"HSA_ENABLE_IPC_MODE_LEGACY": os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"), "MASTER_ADDR": "127.0.0.1"}
    note = "one rank per GPU, RCCL"
    if n_dev < args.gpus:
        if not args.virtual:
            return {"error": "--gpus %d but %d devices visible; --virtual runs the %d ranks on the devices there are (a dry run: "
                             "gloo carries the exchanges)" % (args.gpus, n_dev, args.gpus)}
        env["MI355_BENCH_BACKEND"] = "gloo"
        env["MI355_BENCH_VIRTUAL"] = "1"
        note = "dry run: %d ranks on %d device(s), exchanges over gloo" % (args.gpus, n_dev)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus), "--master-addr",
           "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + [a for a in argv if a not in ("--dry-run", "--virtual")]
    # (--virtual travels in the environment: torch.distributed.run's own parser trips over it -- "ambiguous option")
    return {"cmd": cmd, "env": env, "note": note}


def self_launch(args, argv):
    import subprocess
    plan = launch_plan(args, argv, device_count())
    if args.dry_run:
        print(json.dumps({"launch": plan}))
        return 0
    if "error" in plan:
        raise SystemExit(plan["error"])
    env = dict(os.environ)
    env.update(plan["env"])
    sys.stderr.write("bench.py: %s\n" % plan["note"])
    return subprocess.call(plan["cmd"], env=env)


def live_pmc(args, size, lvl, dominant):
    """The dominant kernel's counters from THIS box, now: three rocprofv3 --pmc passes (SQ counters, FETCH_SIZE, WRITE_SIZE -- the
    TCC counters do not fit one pass; MI355X_MICROARCH.md, HBM section) over one step of this very command.  Per launch; FETCH_SIZE
    doubled as the guide prescribes for gfx950.  None when rocprofv3 is not there or a pass fails (the line then falls back to the
    committed summary and says so)."""
    import csv
    import glob
    import shutil
    import subprocess
    import tempfile
    exe = shutil.which("rocprofv3")
    if not exe or os.environ.get("MI355_BENCH_LIVE_PMC", "1") == "0":
        return None
    passes = [["SQ_INSTS_VALU", "SQ_ACTIVE_INST_VALU", "GRBM_GUI_ACTIVE", "SQ_LDS_BANK_CONFLICT", "SQ_LDS_IDX_ACTIVE"], ["FETCH_SIZE"],
              ["WRITE_SIZE"]]
    tot = {}
    launches = {}
    t0 = time.perf_counter()
    for counters in passes:
        d = tempfile.mkdtemp(prefix="mi355_pmc_", dir="/tmp")
        cmd = [exe, "--pmc"] + counters + ["--output-format", "csv", "-d", d, "--", sys.executable, os.path.abspath(__file__), "--steps", "1",
                                           "--warmup", "0", "--no-cpu-baseline", "--no-host-api", "--no-live-pmc", "--workload", args.workload,
                                           "--size", str(size), "--level", lvl]
        try:
            subprocess.run(cmd, cwd="/tmp", env=dict(os.environ, TMPDIR="/tmp"), stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, timeout=240)
            seen = {}
            for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
                for r in csv.DictReader(open(f)):
                    name = r["Kernel_Name"].replace("void ", "").replace("mi355::", "")
                    if not name.startswith(dominant):  # (k_match3 and k_match3_swz: the walk, whichever table an epoch took)
                        continue
                    tot[r["Counter_Name"]] = tot.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
                    seen.setdefault(r["Counter_Name"], set()).add(r["Dispatch_Id"])
            for cn in counters:
                if cn not in seen:
                    return None
                launches[cn] = len(seen[cn])
        except Exception:
            return None
        finally:
            shutil.rmtree(d, ignore_errors=True)
    out = {"hbm_bytes": int((2 * tot["FETCH_SIZE"] + tot["WRITE_SIZE"]) * 1024), "FETCH_SIZE_KB": tot["FETCH_SIZE"], "WRITE_SIZE_KB": tot["WRITE_SIZE"],
           "SQ_INSTS_VALU": tot["SQ_INSTS_VALU"], "SQ_ACTIVE_INST_VALU": tot["SQ_ACTIVE_INST_VALU"], "GRBM_GUI_ACTIVE": tot["GRBM_GUI_ACTIVE"],
           "lds_bank_conflict_rate": round(tot["SQ_LDS_BANK_CONFLICT"] / max(1.0, tot["SQ_LDS_IDX_ACTIVE"]), 4),
           "valu_wave_instr_per_input_byte": round(tot["SQ_INSTS_VALU"] / size, 3), "dispatches": launches,
           "seconds": round(time.perf_counter() - t0, 1)}
    return out


def committed_digest(workload, total, lvl):
    """length and SHA-256 of the ORACLE's stream of this very input, where one is committed (tests/golden/)"""
    if workload != "webtext" or lvl != "default":
        return None
    for name in ("config5_digest.json", "config5_digest_%d.json" % total):
        p = os.path.join(ROOT, "tests", "golden", name)
        if os.path.isfile(p):
            g = json.load(open(p))["digests"]["raw"]
            if g["in_len"] == total:
                return {"out_len": g["out_len"], "out_sha256": g["out_sha256"], "file": "tests/golden/" + name}
    return None


def stream_check(workload, total, per_rank, world, lvl, out_bytes_fn, out_len):
    """Is the stitched stream the oracle's?  By a committed digest where there is one (config 5 and its smaller totals), by the
    oracle run on the whole input here when that takes seconds, else unknown (None)."""
    import hashlib
    gold = committed_digest(workload, total, lvl)
    if gold is not None:
        h = hashlib.sha256()
        for part in out_bytes_fn():
            h.update(part)
        return {"bit_exact_vs_oracle": bool(out_len == gold["out_len"] and h.hexdigest() == gold["out_sha256"]),
                "oracle_digest": gold["file"], "ref_out_bytes": gold["out_len"]}
    if total <= 512_000_000 and lvl in ("default", "best", "fast"):
        import oracle_binding as ob
        whole = b"".join(make_input(workload, per_rank, r) for r in range(world))
        ref = ob.encode(whole, level={"default": ob.DEFAULT, "best": ob.BEST, "fast": ob.FAST}[lvl])
        got = b"".join(out_bytes_fn())
        return {"bit_exact_vs_oracle": bool(got == ref), "oracle_digest": "the oracle run on the %d bytes here" % total,
                "ref_out_bytes": len(ref)}
    return {"bit_exact_vs_oracle": None, "oracle_digest": None}


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default="", choices=["", "enwik8", "zeros", "random", "silesia", "webtext"],
                    help="default: enwik8 (BASELINE config 3) on one GPU, webtext (config 5: 1 GiB per GPU of one N GiB input) on several")
    ap.add_argument("--size", type=int, default=0, help="bytes per GPU (0 = the config's size)")
    ap.add_argument("--level", default="", choices=["", "default", "best", "fast", "rle", "huffman_only"],
                    help="override the level of the workload (default: Default, rle() for zeros)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-host-api", action="store_true",
                    help="skip the value_host_api leg (profiling runs: every kernel row is then one launch shape)")
    ap.add_argument("--single-process", action="store_true",
                    help="N > 1 without torch.distributed: one process drives all --gpus devices through "
                         "mi355_deflate_encode_multi_device (a thread per device inside the library)")
    ap.add_argument("--virtual", action="store_true",
                    help="N > 1 on a box with fewer GPUs: the ranks share the devices there are (rank r on device r %% devices; "
                         "one process per rank with gloo for the exchanges, or with --single-process inside one call)")
    ap.add_argument("--stitch", default="rccl", choices=["rccl", "peer"],
                    help="--single-process: how the packed ranges reach rank 0's device (MI355_CFG_MULTI_STITCH)")
    ap.add_argument("--dry-run", action="store_true", help="N > 1 without torch.distributed.run: print the launch as JSON and stop")
    ap.add_argument("--pmc-file", default="", help="PMC summary to take roofline.traffic from when no live counters are taken (default: "
                                                    "newest profiles/r*_pmc_summary.json)")
    ap.add_argument("--no-live-pmc", action="store_true",
                    help="do not run the three rocprofv3 --pmc passes that measure roofline.traffic / valu_issue on this box (about a minute)")
    args = ap.parse_args(argv)
    if args.gpus < 1:
        raise SystemExit("--gpus must be at least 1")
    args.virtual = args.virtual or os.environ.get("MI355_BENCH_VIRTUAL") == "1"
    if not args.workload:
        args.workload = "enwik8" if args.gpus == 1 else "webtext"
    launched = "WORLD_SIZE" in os.environ  # (torch.distributed.run sets it, for one process too)
    if args.gpus > 1 and not launched and not args.single_process:
        return self_launch(args, sys.argv[1:] if argv is None else list(argv))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus and not args.single_process:
        raise SystemExit("--gpus %d but torch.distributed.run started %d processes" % (args.gpus, world))
    sizes = {"enwik8": 100_000_000, "zeros": 256 * 1024 * 1024, "random": 64 * 1024 * 1024, "silesia": 212_100_000, "webtext": 1 << 30}
    size = args.size or sizes[args.workload]
    if world > 1 or args.single_process:
        size = (size + 32767) // 32768 * 32768  # rank ranges of the one big input are 32 KiB aligned
    data = None
    if not (args.single_process and args.gpus > 1):
        data = make_input(args.workload, size, rank)  # (before torch and HIP are loaded: a large one is made on a pool of processes)
        if len(data) != size:
            raise SystemExit("workload %s gives %d bytes, not %d" % (args.workload, len(data), size))

    if args.single_process and args.gpus > 1:
        return single_process(args)

    import torch
    import torch.distributed as dist

    import deflate_amd as da
    import shard

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the product has no CPU path")
    # MI355_BENCH_BACKEND=gloo lets several ranks share one GPU (dry run of the N > 1 path on a 1-GPU box):
    # exchanged tensors then travel through host memory; the default is RCCL over xGMI.
    backend = os.environ.get("MI355_BENCH_BACKEND", "nccl")
    if world > torch.cuda.device_count() and backend == "nccl":
        if not args.virtual:
            raise SystemExit("--gpus %d but %d devices visible: RCCL takes one rank per device (--virtual: the ranks share the "
                             "devices and gloo carries the exchanges -- a dry run)" % (world, torch.cuda.device_count()))
        backend = "gloo"
    dev_index = local_rank % torch.cuda.device_count()
    torch.cuda.set_device(dev_index)
    cdev = "cuda" if backend == "nccl" else "cpu"
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        dist.init_process_group(backend, rank=rank, world_size=world)

    shard_mode = os.environ.get("MI355_SHARD_MODE", "p1") if world > 1 else "single"
    lvl = args.level or {"zeros": "rle", "silesia": "best"}.get(args.workload, "default")
    options = {"default": da.CompressionOptions.default, "best": da.CompressionOptions.high,
               "fast": da.CompressionOptions.fast, "rle": da.CompressionOptions.rle,
               "huffman_only": da.CompressionOptions.huffman_only}[lvl]()
    level_name = {"default": "Compression::Default", "best": "Compression::Best", "fast": "Compression::Fast",
                  "rle": "rle()", "huffman_only": "huffman_only()"}[lvl]

    d_in = torch.frombuffer(bytearray(data), dtype=torch.uint8).cuda()
    if world > 1:
        data = None  # (a GiB a rank: the device holds it from here on)
    cap = da.bound(size) + 8
    d_out = torch.empty(cap, dtype=torch.uint8, device="cuda")
    ctx = da.Context(dev_index)
    try:
        ctx.config(da.Context.CFG_STAGE_CLOCKS, 1)  # (kernel_ms and stage_ms come from the call's own events, whatever --size is)
    except da.DeflateError:
        pass  # (a library of before the key -- MI355_DEFLATE_LIB, an A/B run -- has the clocks on anyway)
    ctx.reserve(size + (shard.HISTORY + shard.LOOKAHEAD if world > 1 else 0))  # set-up, like the context itself
    stream = torch.cuda.current_stream().cuda_stream
    flush = shard.flush_mode_for(rank, world)
    total = world * size
    layout = None
    d_ext = None
    if shard_mode == "p1":
        # The one big input is the concatenation of the ranks' shards; a rank also holds 32 KiB of history
        # from its left neighbour and 128 KiB of look-ahead from its right one (exchanged once, untimed).
        layout = shard.p1_layout(total, rank, world)
        parts = []
        if rank > 0:
            prev = torch.empty(shard.HISTORY, dtype=torch.uint8, device=cdev)
        if rank < world - 1:
            nxt = torch.empty(shard.LOOKAHEAD, dtype=torch.uint8, device=cdev)
        # all four transfers of a rank in one group: posted one by one, the sends of neighbouring ranks
        # would wait for each other's receives
        ops = []
        if rank < world - 1:
            ops.append(dist.P2POp(dist.isend, d_in[size - shard.HISTORY:].to(cdev).contiguous(), rank + 1))
            ops.append(dist.P2POp(dist.irecv, nxt, rank + 1))
        if rank > 0:
            ops.append(dist.P2POp(dist.isend, d_in[: shard.LOOKAHEAD].to(cdev).contiguous(), rank - 1))
            ops.append(dist.P2POp(dist.irecv, prev, rank - 1))
        for q in dist.batch_isend_irecv(ops):
            q.wait()
        if rank > 0:
            parts.append(prev.cuda())
        parts.append(d_in)
        if rank < world - 1:
            parts.append(nxt.cuda())
        # open the other connections the timed steps use (every rank -> rank 0 for the stitch, all-gather), so
        # that the lazy set-up of RCCL's channels is not timed even with --warmup 0
        one = torch.zeros(1, dtype=torch.uint8, device=cdev)
        if rank == 0:
            tmp = [torch.zeros(1, dtype=torch.uint8, device=cdev) for _ in range(world - 1)]
            ops = [dist.P2POp(dist.irecv, tmp[r - 1], r) for r in range(1, world)]
        else:
            ops = [dist.P2POp(dist.isend, one, 0)]
        for q in dist.batch_isend_irecv(ops):
            q.wait()
        dist.all_gather([torch.zeros(1, dtype=torch.int64, device=cdev) for _ in range(world)],
                        torch.zeros(1, dtype=torch.int64, device=cdev))
        parts.append(torch.zeros(64, dtype=torch.uint8, device="cuda"))
        d_ext = torch.cat(parts)
        assert d_ext.numel() - 64 == layout["g_hi"] - layout["g_lo"]

    out_len = [0]
    last_img = [None]
    match_ms = []
    stage_ms = {}
    gpu_ms = []

    def step(record):
        if shard_mode == "p1":
            img, n = shard.encode_p1_dist(da, ctx, d_ext, layout, total, rank, world, options, comm_device=cdev)
            out_len[0] = n if rank == 0 else 0
            last_img[0] = img  # (rank 0: a view of the stitched stream, valid until the next step)
            if record:
                info = ctx.info()
                match_ms.append(info["match_ms"] / max(1, info["match_launches"]))
                gpu_ms.append(info["total_ms"])
            return
        n = ctx.encode_device(d_in.data_ptr(), size, d_out.data_ptr(), cap, options, stream=stream, flush=flush)
        out_len[0] = n
        if record:
            info = ctx.info()
            match_ms.append(info["match_ms"] / max(1, info["match_launches"]))
            gpu_ms.append(info["total_ms"])
            for k, v in info["stage_ms"].items():
                stage_ms[k] = stage_ms.get(k, 0.0) + v
        if world > 1:
            shard.stitch(d_out if cdev == "cuda" else d_out.cpu(), n, rank, world)

    for _ in range(args.warmup):
        step(False)
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step(True)
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    # (N = 1: the same steps once more as a caller's context runs them -- without the per-stage clocks this bench asks for, which
    # are five events, 29 us of idle queue, in every step above: value_as_called)
    elapsed_plain = None
    if world == 1:
        try:
            ctx.config(da.Context.CFG_STAGE_CLOCKS, 0)
            step(False)
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            for _ in range(args.steps):
                step(False)
            torch.cuda.synchronize()
            elapsed_plain = time.perf_counter() - t1
            ctx.config(da.Context.CFG_STAGE_CLOCKS, 1)
        except da.DeflateError:
            elapsed_plain = None
    p1_trace = None
    rccl_ranks = 0
    if shard_mode == "p1":  # one more step, untimed, with a device synchronisation after every phase
        img, _ = shard.encode_p1_dist(da, ctx, d_ext, layout, total, rank, world, options, comm_device=cdev, trace=True)
        last_img[0] = img
        p1_trace = dict(shard.LAST_TRACE)
    if world > 1 and backend == "nccl":
        # what RCCL itself saw: a sum of ones over the communicator the steps used, on the devices
        ones = torch.ones(1, dtype=torch.int32, device="cuda")
        dist.all_reduce(ones)
        rccl_ranks = int(ones.item())
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=cdev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        tot_out = torch.tensor([out_len[0]], dtype=torch.int64, device=cdev)
        dist.all_reduce(tot_out)
        total_out = int(tot_out.item())  # p1: only rank 0 reports the (whole) stream length
    else:
        total_out = out_len[0]

    if rank == 0:
        ms_per_step = elapsed * 1e3 / args.steps
        value = world * size * args.steps / elapsed / 1e6
        mm = sum(match_ms) / len(match_ms)
        # every level with a hash budget runs k_sort + k_match3 (Fast included); rle() runs k_rle; huffman_only() has no
        # match stage (the table is a fill)
        sorted_walk = lvl in ("default", "best", "fast")
        dominant = "k_rle" if lvl == "rle" else ("k_match3" if sorted_walk else "fill")
        # 1 B read + r B written per input byte; one launch of the dominant kernel = one rank's bytes
        algo_bytes = size + (total_out // world)
        achieved = algo_bytes / (mm * 1e-3) / 1e9 if mm > 0 else 0.0
        # HBM bytes per launch of the dominant kernel and the LDS bank-conflict rate of the match compare, from
        # the committed PMC passes of this very command (profiles/, FETCH_SIZE doubled as the guide prescribes)
        traffic = None
        lds_conflict = None
        pmc_source = None
        valu_issue = None
        live = None
        if world == 1 and not args.no_live_pmc and dominant in ("k_match3", "k_rle"):
            live = live_pmc(args, size, lvl, dominant)
        if live is not None:
            traffic = live["hbm_bytes"]
            lds_conflict = live["lds_bank_conflict_rate"]
            pmc_source = ("live: three rocprofv3 --pmc passes over one step of this command on this box, %.0f s (FETCH_SIZE x 2 + WRITE_SIZE; "
                          "%s launches of the kernel per pass)" % (live["seconds"], sorted(set(live["dispatches"].values()))))
            valu_issue = {"wave_instr_per_input_byte": live["valu_wave_instr_per_input_byte"],
                          "busy": round(live["SQ_ACTIVE_INST_VALU"] * 4.0 / 1024.0 / (live["GRBM_GUI_ACTIVE"] / 8.0), 3)}
        else:
          try:
              import glob
              cands = [args.pmc_file] if args.pmc_file else sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_summary.json")))
              pm = json.load(open(cands[-1]))
              k = pm["kernels"][dominant]
              # a summary is only taken when it is of this very workload and every counter of the kernel was
              # averaged over launches of ONE shape: launches == steps of the profiled command (the host-API leg,
              # which launches the kernel on pieces of the input, must have been off)
              if (pm["workload"] == args.workload and pm["bytes_per_gpu"] == size and pm["level"] == lvl and world == 1
                      and pm.get("launches_per_pass") == k.get("launches") and k.get("launches")):
                  traffic = k["hbm_bytes"]
                  lds_conflict = k.get("lds_bank_conflict_rate")
                  pmc_source = os.path.relpath(cands[-1], ROOT)
                  # what the kernel is really bound by (the contract's "bound" knows hbm and mfma only): vector-ALU issue.
                  # A wave64 instruction holds its SIMD for four cycles; busy = issued wave instructions x 4 over the cycles
                  # of the chip's 1024 SIMDs (GRBM_GUI_ACTIVE is summed over the 8 XCDs)
                  if k.get("SQ_ACTIVE_INST_VALU") and k.get("GRBM_GUI_ACTIVE"):
                      valu_issue = {"wave_instr_per_input_byte": k.get("valu_wave_instr_per_input_byte"),
                                    "busy": round(k["SQ_ACTIVE_INST_VALU"] * 4.0 / 1024.0 / (k["GRBM_GUI_ACTIVE"] / 8.0), 3)}
          except Exception:
              pass
        # the kernel that reads the input coalesced: k_sort files every position under its hash (reads n, writes
        # the sorted array and the bucket starts, 2 B per position each); on the unsorted path k_links_a
        links_ms = stage_ms.get("links", 0.0) / args.steps if stage_ms else 0.0
        in_kernel = "k_sort" if sorted_walk else "k_links_a+k_links_b"
        in_bytes = size * 5
        input_load = None
        if links_ms > 0 and lvl not in ("rle", "huffman_only"):
            # twice: the input bytes alone (no credit for intermediates), and with the sorted
            # array and bucket starts the kernel writes (2 B per position each)
            input_load = {"kernel": in_kernel, "ms": round(links_ms, 3), "unit": "GB/s",
                          "input_only": {"bytes": size, "achieved": round(size / (links_ms * 1e-3) / 1e9, 1),
                                         "frac": round(size / (links_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)},
                          "with_intermediates": {"bytes": in_bytes, "achieved": round(in_bytes / (links_ms * 1e-3) / 1e9, 1),
                                                 "frac": round(in_bytes / (links_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)}}
        res = {
            "metric": "MB/s raw input encoded (%s) + compressed size vs ref" % level_name,
            "value": round(value, 2), "unit": "MB/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(ms_per_step, 3), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "u8", "data": "real" if "real" in INPUT_NOTE.get(args.workload, "") else "synthetic",
            "config": {"workload": "%s%s: %d bytes per GPU, %s, %s" % (
                args.workload, " = " + INPUT_NOTE[args.workload] if args.workload in INPUT_NOTE else "", size, level_name,
                "stream-exact (P1)" if world == 1 else (
                    "one %d-byte input sharded over %d GPUs, stream-exact (P1), stitch over %s" % (
                        total, world, "RCCL" if backend == "nccl" else backend + " (dry run: the ranks may share a GPU)")
                    if shard_mode == "p1" else "chunk-exact (P2) stitch across GPUs")),
                "bytes_per_gpu": size, "level": level_name, "parallelism": "shard%d" % world},
            "out_bytes": total_out, "ratio": round(total_out / (world * size), 5),
            "gpu_ms_per_step_events": round(sum(gpu_ms) / len(gpu_ms), 3),
            # every timed step by its own HIP events (first kernel to last): one noisy step shows here, not only in the mean
            "step_ms_events": {"min": round(min(gpu_ms), 3), "median": round(sorted(gpu_ms)[len(gpu_ms) // 2], 3),
                               "max": round(max(gpu_ms), 3), "n": len(gpu_ms)},
            "stage_ms": {k: round(v / args.steps, 3) for k, v in stage_ms.items()},
            # the same steps as a caller's context runs them: without the per-stage clocks (five events in every step above)
            "value_as_called": ({"value": round(size * args.steps / elapsed_plain / 1e6, 2), "unit": "MB/s",
                                 "ms_per_step": round(elapsed_plain * 1e3 / args.steps, 3),
                                 "what": "the timed steps once more with MI355_CFG_STAGE_CLOCKS = 0, a context's default"}
                                if elapsed_plain else None),
            "roofline": {"bound": "hbm", "kernel": dominant, "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS,
                         "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 5), "traffic": traffic,
                         "kernel_ms": round(mm, 3), "algorithmic_bytes_per_launch": algo_bytes,
                         "input_load": input_load, "lds_bank_conflict_rate": lds_conflict, "valu_issue": valu_issue,
                         "pmc_source": pmc_source},
        }
        if p1_trace is not None:
            res["p1_phases_ms_rank0"] = p1_trace  # (one untimed step, synchronised per phase)
        if world > 1:
            res["rccl_ranks"] = rccl_ranks  # (0: the exchanges did not travel over RCCL -- the gloo dry run)
            res["stitch"] = "rccl" if backend == "nccl" else backend
            res["devices"] = torch.cuda.device_count()
            if shard_mode == "p1" and last_img[0] is not None:
                img = last_img[0]

                def parts():
                    for i in range(0, total_out, 256 << 20):
                        yield bytes(img[i:min(total_out, i + (256 << 20))].cpu().numpy())
                res.update(stream_check(args.workload, total, size, world, lvl, parts, total_out))
        if world == 1 and not args.no_host_api:
            # the drop-in call itself, deflate_bytes(&[u8]) -> Vec<u8> (src/lib.rs:163): pinned host buffers,
            # first H2D byte to last D2H byte inside the timed region
            h_in = torch.frombuffer(bytearray(data), dtype=torch.uint8).pin_memory()
            h_out = torch.empty(cap + 64, dtype=torch.uint8).pin_memory()
            ctx.reserve(size, host_api=True)
            ctx.encode_host_ptr(h_in.data_ptr(), size, h_out.data_ptr(), cap + 64, options)
            reps = max(2, args.steps)
            calls = []
            t1 = time.perf_counter()
            for _ in range(reps):
                tc = time.perf_counter()
                hn = ctx.encode_host_ptr(h_in.data_ptr(), size, h_out.data_ptr(), cap + 64, options)
                calls.append((time.perf_counter() - tc) * 1e3)  # (the call returns when the last byte has landed)
            dt = time.perf_counter() - t1
            res["value_host_api"] = {"value": round(size * reps / dt / 1e6, 2), "unit": "MB/s",
                                     "ms_per_call": round(dt * 1e3 / reps, 3),
                                     "call_ms": {"min": round(min(calls), 3), "median": round(sorted(calls)[len(calls) // 2], 3),
                                                 "max": round(max(calls), 3), "n": len(calls)},
                                     "what": "mi355_deflate_encode on pinned host buffers, first H2D byte to last D2H byte (pieces: H2D, "
                                             "kernels and the copy engine's D2H overlap)",
                                     "same_bytes": bool(hn == out_len[0] and torch.equal(
                                         h_out[:hn], d_out[:hn].cpu()))}
        if world == 1 and not args.no_host_api:
            # ... and on the memory a drop-in caller has (src/lib.rs:137-147: &[u8] in, Vec<u8> out): plain pageable buffers.
            # The context's host threads carry them through page-locked slots (MI355_CFG_HOST_BOUNCE, deflate_bounce.inc);
            # beside it the same call with the threads off -- the runtime's own copies, one after the other
            import ctypes
            import numpy as np
            p_in = np.frombuffer(data, dtype=np.uint8).copy()
            p_out = np.zeros(cap + 64, dtype=np.uint8)  # (touched: a fresh Vec's page faults are the caller's in any implementation)
            reps = max(2, args.steps)

            def pageable(bounce):
                ctx.config(da.Context.CFG_HOST_BOUNCE, bounce)
                for _ in range(3):  # (untimed: the first call of a kind makes the context's host threads and their rings)
                    hn = ctx.encode_host_ptr(p_in.ctypes.data, size, p_out.ctypes.data, cap + 64, options)
                calls = []
                t1 = time.perf_counter()
                for _ in range(reps):
                    tc = time.perf_counter()
                    hn = ctx.encode_host_ptr(p_in.ctypes.data, size, p_out.ctypes.data, cap + 64, options)
                    calls.append((time.perf_counter() - tc) * 1e3)
                dt = time.perf_counter() - t1
                return {"value": round(size * reps / dt / 1e6, 2), "unit": "MB/s", "ms_per_call": round(dt * 1e3 / reps, 3),
                        "call_ms": {"min": round(min(calls), 3), "median": round(sorted(calls)[len(calls) // 2], 3),
                                    "max": round(max(calls), 3), "n": len(calls)},
                        "host_path": ctx.info()["host_path"],
                        "same_bytes": bool(hn == out_len[0] and np.array_equal(p_out[:hn], d_out[:hn].cpu().numpy()))}
            slow = pageable(0)
            fast = pageable(1)
            fast["what"] = ("mi355_deflate_encode on PAGEABLE host buffers (numpy arrays: what a &[u8] / Vec<u8> caller hands over), "
                            "first byte in to last byte out; host_path bits: 1 pieces, 2 input by the context's host threads, 4 output by them")
            fast["runtime_copies"] = slow  # (MI355_CFG_HOST_BOUNCE = 0: what such a caller got before)
            res["value_host_api_pageable"] = fast
        if world == 1 and not args.no_host_api:
            # ... and what a SMALL call costs (the reference's own consumers -- a PNG's rows -- live there): the reference's
            # fixture pg11.txt, 167 KB, resident and through the host-buffer call on pageable memory, wall clock, the
            # context as a caller gets it (no per-stage events: MI355_CFG_STAGE_CLOCKS)
            pg = os.path.join(ROOT, "tests", "golden", "ref_inputs", "pg11.txt")
            if os.path.exists(pg):
                import ctypes
                small = open(pg, "rb").read()
                sctx = da.Context(dev_index)
                s_in = torch.frombuffer(bytearray(small), dtype=torch.uint8).cuda()
                s_cap = da.bound(len(small)) + 16
                s_out = torch.empty(s_cap, dtype=torch.uint8, device="cuda")
                h_in = (ctypes.c_uint8 * len(small)).from_buffer_copy(small)
                h_o = (ctypes.c_uint8 * s_cap)()

                def med(fn, reps=60):
                    for _ in range(5):
                        fn()
                    ts = []
                    for _ in range(reps):
                        torch.cuda.synchronize()
                        tc = time.perf_counter()
                        fn()
                        ts.append((time.perf_counter() - tc) * 1e3)
                    ts.sort()
                    return round(ts[len(ts) // 2], 4), round(ts[0], 4)
                sn = [0]
                r_med, r_min = med(lambda: sn.__setitem__(0, sctx.encode_device(s_in.data_ptr(), len(small), s_out.data_ptr(), s_cap, options)))
                h_med, h_min = med(lambda: sctx.encode_host_ptr(ctypes.addressof(h_in), len(small), ctypes.addressof(h_o), s_cap, options))
                res["small_call"] = {"input": "tests/golden/ref_inputs/pg11.txt", "bytes": len(small), "out_bytes": sn[0],
                                     "resident_ms": {"median": r_med, "min": r_min},
                                     "host_call_pageable_ms": {"median": h_med, "min": h_min},
                                     "same_bytes": bool(bytes(h_o[:sn[0]]) == bytes(s_out[:sn[0]].cpu().numpy()))}
                small_got = bytes(h_o[:sn[0]])
                sctx.close()
        if world == 1 and not args.no_cpu_baseline:
            import oracle_binding as ob
            olvl = {"default": ob.DEFAULT, "best": ob.BEST, "fast": ob.FAST, "rle": ob.RLE,
                    "huffman_only": ob.HUFFMAN_ONLY}[lvl]
            res["cpu_baseline"] = cpu_baselines(data, olvl, level_name)
            # the whole workload once more through the oracle: the stream the GPU produced must be its stream
            t1 = time.perf_counter()
            ref = ob.encode(data, level=olvl)
            res["cpu_baseline"]["whole_workload_s"] = round(time.perf_counter() - t1, 2)
            got = bytes(d_out[: out_len[0]].cpu().numpy())
            res["ref_out_bytes"] = len(ref)
            res["bit_exact_vs_oracle"] = bool(got == ref)
            if "small_call" in res:
                res["small_call"]["bit_exact_vs_oracle"] = bool(small_got == ob.encode(small, level=olvl))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        print(json.dumps(res))
    quiet_stdout()


if __name__ == "__main__":
    sys.exit(main() or 0)
