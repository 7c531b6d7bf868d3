#!/bin/bash
# development aid (GPU box): the bench line's value and stage clocks for the product and every deflate-rs_amd/variants/v_*.so, three times round
R=$GRAFT_REPO_ROOT
for round in 1 2 3; do
for f in $R/deflate-rs_amd/libmi355deflate.so $R/deflate-rs_amd/variants/v_*.so; do
MI355_DEFLATE_LIB=$f timeout -s KILL 300 python $R/bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-host-api --no-live-pmc "$@" 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('%-24s' % sys.argv[1], d['value'], d['ms_per_step'], d['stage_ms'], d.get('bit_exact_vs_oracle'))" $(basename $f)
done
done
