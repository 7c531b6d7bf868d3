#!/bin/bash
# A/B of an environment switch on the bench line's numbers
for v in "$@"; do
  env $v python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-host-api 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('%-28s' % '$v', d['value'], d['ms_per_step'], d['stage_ms'], d.get('bit_exact_vs_oracle'))"
done
