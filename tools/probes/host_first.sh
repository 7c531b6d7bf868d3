#!/bin/bash
# development aid (GPU box): the host call with first pieces of different sizes (MI355_HOST_FIRST: epochs; 0 = a whole round)
for f in 0 64 128 32 0 64 96; do
  MI355_HOST_FIRST=$f python bench.py --steps 12 --warmup 1 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); h=d['value_host_api']; print('first %-6s' % '$f', h['value'], h['call_ms'], h['same_bytes'])"
done
