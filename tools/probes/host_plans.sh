#!/bin/bash
# development aid (GPU box): the host call with the pieces of its input planned by hand (MI355_HOST_PLAN: rounds of 256 epochs per piece,
# behind the quarter-round first piece)
for plan in "" "1,2,3,3,2" "1,2,3,2,2,1" "1,2,2,2,2,2" "" "1,2,3,4,1" "1,3,3,3,1" "$@"; do
  MI355_HOST_PLAN=$plan python bench.py --steps 12 --warmup 1 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); h=d['value_host_api']; print('%-18s' % '$plan', h['value'], h['call_ms'], h['same_bytes'])"
done
