"""Sum rocprofv3 --pmc counters per kernel from the counter_collection CSVs under a directory.
usage: python tools/pmc_summary.py DIR [kernel-substring]  -> JSON {kernel: {counter: per-launch value, "launches": n}}"""
import csv, glob, json, os, sys
from collections import defaultdict
acc = defaultdict(lambda: defaultdict(float)); launches = defaultdict(set)
for f in glob.glob(os.path.join(sys.argv[1], "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0].replace("mi355::", "").replace("void ", "")
        if len(sys.argv) > 2 and sys.argv[2] not in k:
            continue
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
        launches[(k, r["Counter_Name"])].add(r["Dispatch_Id"])
out = {}
for k, cs in acc.items():
    out[k] = {c: v / max(1, len(launches[(k, c)])) for c, v in cs.items()}
    out[k]["launches"] = max(len(launches[(k, c)]) for c in cs)
print(json.dumps(out, indent=1, sort_keys=True))
