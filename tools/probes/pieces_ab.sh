#!/bin/bash
# development aid (GPU box): the device-resident call in pieces (rounds of workgroups per piece), block stages beside the next walk
for v in "X=0" "MI355_PIECES=3,3,3,3" "MI355_PIECES=3,3,3,2,1" "MI355_PIECES=2,2,2,2,2,2" "MI355_PIECES=4,4,3,1" "MI355_PIECES=6,5,1" "MI355_PIECES=3,3,3,3 MI355_TWO_STREAMS=0" "X=0"; do
  env $v python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-host-api 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('%-44s' % '$v', d['value'], d['ms_per_step'], d.get('bit_exact_vs_oracle'))"
done
