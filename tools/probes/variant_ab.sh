#!/bin/bash
# development aid (GPU box): k_match3's time and vector instructions for the product and every deflate-rs_amd/variants/v_*.so on one input
#   variant_ab.sh records96 best
R=$GRAFT_REPO_ROOT; cd /tmp && export TMPDIR=/tmp
for f in $R/deflate-rs_amd/libmi355deflate.so $R/deflate-rs_amd/variants/v_*.so; do
rm -rf /tmp/vab /tmp/vabp
MI355_DEFLATE_LIB=$f timeout -s KILL 200 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/vab -- python $R/tools/probes/loop_any.py $1 $2 3 ${3:-20} > /tmp/vab.log 2>&1
MI355_DEFLATE_LIB=$f timeout -s KILL 200 rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_INSTS_SALU GRBM_GUI_ACTIVE --output-format csv -d /tmp/vabp -- python $R/tools/probes/loop_any.py $1 $2 3 ${3:-20} > /tmp/vabp.log 2>&1
python - "$f" <<'PY'
import csv, glob, sys
name = sys.argv[1].split("/")[-1]
f = glob.glob("/tmp/vab/**/*kernel_stats.csv", recursive=True)
for r in csv.DictReader(open(f[0])) if f else []:
    if "k_match3" in r["Name"]:
        print("%-24s %-14s calls %s avg %9.1f us" % (name, "k_match3_swz" if "k_match3_swz" in r["Name"] else "k_match3", r["Calls"], float(r["AverageNs"]) / 1e3))
tot = {}
for g in glob.glob("/tmp/vabp/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(g)):
        if "k_match3" in r["Kernel_Name"]:
            tot[r["Counter_Name"]] = tot.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
n = 3
if tot:
    print("%-24s per launch: %s" % (name, "  ".join("%s %.3g" % (k, v / n) for k, v in sorted(tot.items()))))
    if "SQ_ACTIVE_INST_VALU" in tot and "GRBM_GUI_ACTIVE" in tot:
        print("%-24s valu busy %.3f" % (name, tot["SQ_ACTIVE_INST_VALU"] * 4 / 1024 / (tot["GRBM_GUI_ACTIVE"] / 8)))
PY
done
