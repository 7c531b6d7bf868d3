#!/bin/bash
# development aid (GPU box): the host-buffer call with compute units kept away from the walk (MI355_RESERVE_CUS)
for v in "MI355_RESERVE_CUS=0" "MI355_RESERVE_CUS=8" "MI355_RESERVE_CUS=16" "MI355_RESERVE_CUS=32" "MI355_RESERVE_CUS=8 MI355_BLK_PRIO=0" "MI355_RESERVE_CUS=0"; do
  env $v python bench.py --steps 5 --warmup 2 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$v', d['value'], d['value_host_api'])"
done
MI355_RESERVE_CUS=8 bash tools/host_api_timeline.sh 2>&1 | tail -52
