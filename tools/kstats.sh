#!/bin/bash
# development aid (GPU box): per-kernel average durations of a short bench run -> gpurun_out/kstats_$1.txt
tag=${1:-x}; shift
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ks_$tag -o ks -- python $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-host-api "$@" > /tmp/ks_$tag.json 2>/tmp/ks_$tag.err
cd $GRAFT_REPO_ROOT
f=$(find /tmp/ks_$tag -name '*kernel_stats.csv' | head -1)
python - "$f" <<'PY' > gpurun_out/kstats_$tag.txt
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
for r in rows[:16]:
    print('%-34s %5s %10.1f' % (r['Name'][:34], r['Calls'], float(r['AverageNs'])/1e3))
PY
cat gpurun_out/kstats_$tag.txt; python -c "
import json;d=json.loads(open('/tmp/ks_$tag.json').read());print(d['value'],d['ms_per_step'],d['stage_ms'])"
