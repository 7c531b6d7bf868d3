#!/bin/bash
# development aid (GPU box): k_match3's LDS bank-conflict share and LDS busy share on one input
#   lds_conflicts.sh records96 best
R=$GRAFT_REPO_ROOT; cd /tmp && export TMPDIR=/tmp
for kind in "$@"; do
rm -rf /tmp/ldc
timeout -s KILL 200 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d /tmp/ldc -- python $R/tools/probes/loop_any.py $kind ${LEVEL:-best} 3 20 > /tmp/ldc.log 2>&1
python - "$kind" <<'PY'
import csv, glob, sys
tot = {}
for g in glob.glob("/tmp/ldc/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(g)):
        if "k_match3" in r["Kernel_Name"]:
            tot[r["Counter_Name"]] = tot.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
if tot:
    print("%-12s conflict/idx_active %.3f   lds idx_active / gui_active(per xcd) %.3f   LDS instr %.3g  VALU instr %.3g" % (
        sys.argv[1], tot["SQ_LDS_BANK_CONFLICT"] / max(1, tot["SQ_LDS_IDX_ACTIVE"]),
        tot["SQ_LDS_IDX_ACTIVE"] / max(1.0, tot["GRBM_GUI_ACTIVE"]) , tot["SQ_INSTS_LDS"] / 3, tot["SQ_INSTS_VALU"] / 3), {k: "%.3g" % (v / 3) for k, v in tot.items()})
PY
done
