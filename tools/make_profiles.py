"""Run on the GPU box from the repo root: the rocprofv3 evidence behind the bench line.
  python tools/make_profiles.py TAG        -> gpurun_out/TAG_kernel_stats.txt, TAG_pmc_summary.json, TAG_bench.json
(copy them into profiles/).  Kernel trace and the counter passes are separate runs of the same command,
`python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-host-api` (trace) / `--steps 1 --warmup 0` (counters:
one launch of every kernel, all of one shape -- the host-API leg, which launches the match kernels on pieces of
the input, is off; the summary records the launch count per kernel and bench.py refuses a file where it differs);
FETCH_SIZE and WRITE_SIZE in passes of their own, FETCH_SIZE doubled (MI355X_MICROARCH.md, HBM section)."""
import csv
import glob
import json
import os
import subprocess
import sys
from collections import defaultdict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TAG = sys.argv[1] if len(sys.argv) > 1 else "r02"
EXTRA = sys.argv[2:]
OUT = os.path.join(ROOT, "gpurun_out")
ENV = dict(os.environ, TMPDIR="/tmp")
LAUNCHES = {}  # kernel -> launches per counter pass


def short(name):
    n = name.replace("void ", "").replace("mi355::", "")
    n = n.split("(")[0]
    if n.startswith("k_emit<"):  # (three kernels: 0 = entries given, 1 = speculative entries, 2 = the repair)
        return n.split(">")[0] + ">"
    return n.split("<")[0] if n.startswith("k_") else n


def rocprof(args, sub):
    d = os.path.join(OUT, "%s_%s" % (TAG, sub))
    subprocess.run("rm -rf " + d, shell=True)
    cmd = ["rocprofv3"] + args + ["--output-format", "csv", "-d", d, "--", sys.executable, os.path.join(ROOT, "bench.py"),
                                  "--no-cpu-baseline", "--no-host-api", "--no-live-pmc"] + EXTRA
    subprocess.run(cmd, cwd="/tmp", env=ENV, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, timeout=600)
    return d


def pmc(counters, steps_args):
    d = rocprof(["--pmc"] + counters, "pmc")
    acc = defaultdict(lambda: defaultdict(float))
    seen = defaultdict(set)
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            k = short(r["Kernel_Name"])
            acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
            seen[(k, r["Counter_Name"])].add(r["Dispatch_Id"])
    subprocess.run("rm -rf " + d, shell=True)
    for (k, cn), ids in seen.items():
        LAUNCHES[k] = max(LAUNCHES.get(k, 0), len(ids))
    return {k: {c: v / max(1, len(seen[(k, c)])) for c, v in cs.items()} for k, cs in acc.items()}


def main():
    global EXTRA
    base_extra = list(EXTRA)
    # 1. kernel trace
    EXTRA = base_extra + ["--steps", "5", "--warmup", "2"]
    d = rocprof(["--kernel-trace", "--stats"], "trace")
    f = glob.glob(os.path.join(d, "**", "*kernel_stats.csv"), recursive=True)[0]
    rows = list(csv.DictReader(open(f)))
    with open(os.path.join(OUT, TAG + "_kernel_stats.txt"), "w") as o:
        o.write("# rocprofv3 --kernel-trace --stats -- python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-host-api %s\n" % " ".join(base_extra))
        o.write("%-28s %6s %12s %12s %12s %8s\n" % ("kernel", "calls", "avg_us", "min_us", "max_us", "pct"))
        for r in rows:
            o.write("%-28s %6s %12.1f %12.1f %12.1f %8.2f\n" % (short(r["Name"])[:28], r["Calls"], float(r["AverageNs"]) / 1e3,
                                                               float(r["MinNs"]) / 1e3, float(r["MaxNs"]) / 1e3, float(r["Percentage"])))
    stats = {short(r["Name"]): float(r["AverageNs"]) / 1e3 for r in rows}
    subprocess.run("rm -rf " + d, shell=True)
    # 2. counters (one launch of every kernel per run)
    EXTRA = base_extra + ["--steps", "1", "--warmup", "0"]
    a = pmc(["SQ_WAVE_CYCLES", "SQ_BUSY_CYCLES", "SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS", "SQ_LDS_BANK_CONFLICT",
             "SQ_LDS_IDX_ACTIVE", "SQ_INSTS_VMEM_RD"], None)
    b = pmc(["SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_ACTIVE_INST_VALU", "TA_BUSY_avr", "GRBM_GUI_ACTIVE"], None)
    fe = pmc(["FETCH_SIZE"], None)
    wr = pmc(["WRITE_SIZE"], None)
    # the bench line of an unprofiled run
    EXTRA = base_extra + ["--steps", "5", "--warmup", "2"]
    line = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--no-live-pmc", "--pmc-file", "/nonexistent"] + EXTRA, capture_output=True, text=True, timeout=600).stdout.strip().splitlines()[-1]
    open(os.path.join(OUT, TAG + "_bench.json"), "w").write(line + "\n")
    bl = json.loads(line)
    n = bl["config"]["bytes_per_gpu"]
    kernels = {}
    for k in sorted(set(a) | set(fe) | set(wr)):
        e = dict(a.get(k, {}))
        e.update(b.get(k, {}))
        if k in fe:
            e["FETCH_SIZE_KB"] = fe[k].get("FETCH_SIZE", 0.0)
        if k in wr:
            e["WRITE_SIZE_KB"] = wr[k].get("WRITE_SIZE", 0.0)
        e["hbm_bytes"] = int((2 * e.get("FETCH_SIZE_KB", 0.0) + e.get("WRITE_SIZE_KB", 0.0)) * 1024)
        if e.get("SQ_LDS_IDX_ACTIVE"):
            e["lds_bank_conflict_rate"] = round(e.get("SQ_LDS_BANK_CONFLICT", 0.0) / e["SQ_LDS_IDX_ACTIVE"], 4)
        if "SQ_INSTS_VALU" in e:
            e["valu_wave_instr_per_input_byte"] = round(e["SQ_INSTS_VALU"] / n, 3)
            e["salu_wave_instr_per_input_byte"] = round(e.get("SQ_INSTS_SALU", 0.0) / n, 3)
        e["launches"] = LAUNCHES.get(k, 0)
        if k in stats:
            e["avg_us_kernel_trace"] = round(stats[k], 1)
            if e["hbm_bytes"]:
                e["hbm_GBps"] = round(e["hbm_bytes"] / (stats[k] * 1e-6) / 1e9, 1)
        kernels[k] = e
    lvl = bl["config"]["level"]
    json.dump({"note": "rocprofv3 --pmc passes (SQ set A, SQ set B, FETCH_SIZE, WRITE_SIZE: four separate runs) over `python bench.py "
                       "--steps 1 --warmup 0 --no-cpu-baseline --no-host-api`; per launch (launches = dispatches of the kernel in a pass); hbm_bytes = (2 * FETCH_SIZE + WRITE_SIZE) * 1024 "
                       "(FETCH_SIZE doubled per MI355X_MICROARCH.md); durations from the --kernel-trace run",
               "workload": bl["config"]["workload"].split(":")[0].split(" = ")[0], "bytes_per_gpu": n,
               "level": {"Compression::Default": "default", "Compression::Best": "best", "Compression::Fast": "fast", "rle()": "rle",
                         "huffman_only()": "huffman_only"}[lvl],
               "launches_per_pass": 1, "kernels": kernels}, open(os.path.join(OUT, TAG + "_pmc_summary.json"), "w"), indent=1, sort_keys=True)
    print(open(os.path.join(OUT, TAG + "_kernel_stats.txt")).read())
    print(json.dumps({k: {x: kernels[k].get(x) for x in ("hbm_bytes", "hbm_GBps", "lds_bank_conflict_rate", "valu_wave_instr_per_input_byte", "avg_us_kernel_trace")} for k in ("k_match3", "k_sort", "k_adv", "k_emit<1>", "k_pack", "k_compact") if k in kernels}, indent=1))


main()
