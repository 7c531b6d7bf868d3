#!/bin/bash
# development aid (GPU box): per-kernel average times (rocprofv3 --stats) of tools/probes/resident_loop.py for each build
# variant under deflate-rs_amd/variants/ -- kernel_ab.sh PATTERN prints the kernels whose name matches
cd /tmp && export TMPDIR=/tmp
for f in $GRAFT_REPO_ROOT/deflate-rs_amd/libmi355deflate.so $GRAFT_REPO_ROOT/deflate-rs_amd/variants/v_*.so; do
rm -rf /tmp/kab; MI355_DEFLATE_LIB=$f rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kab -- python $GRAFT_REPO_ROOT/tools/probes/resident_loop.py 5 > /tmp/kab.log 2>&1
python - "$f" "${1:-k_}" <<'PY'
import csv, glob, sys
f = glob.glob("/tmp/kab/**/*kernel_stats.csv", recursive=True)[0]
for r in csv.DictReader(open(f)):
    if sys.argv[2] in r["Name"]:
        print("%-28s %-36s calls %3s avg %9.1f us" % (sys.argv[1].split("/")[-1], r["Name"].split("(")[0][-36:], r["Calls"], float(r["AverageNs"]) / 1e3))
PY
done
