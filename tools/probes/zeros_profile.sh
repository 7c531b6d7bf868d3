cd /tmp && export TMPDIR=/tmp
for lv in "" "--level default"; do
rm -rf /tmp/zp; rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/zp -- python $GRAFT_REPO_ROOT/bench.py --workload zeros $lv --steps 4 --warmup 2 --no-cpu-baseline --no-host-api > /tmp/zp.log 2>&1
python - <<'PY'
import csv, glob
f = glob.glob("/tmp/zp/**/*kernel_stats.csv", recursive=True)[0]
for r in list(csv.DictReader(open(f)))[:12]:
    print("%-40s calls %4s avg %9.1f us  pct %5s" % (r["Name"].split("(")[0][-40:], r["Calls"], float(r["AverageNs"])/1e3, r["Percentage"]))
PY
done
