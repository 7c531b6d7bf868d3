#!/bin/bash
# development aid (GPU box): the host call's rate for each build variant under deflate-rs_amd/variants/
for r in 1 2; do
for f in deflate-rs_amd/variants/v_*.so; do
  MI355_DEFLATE_LIB=$PWD/$f timeout 200 python bench.py --steps 3 --warmup 1 --no-cpu-baseline "$@" 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); h=d['value_host_api']; print('%-40s' % '$f', d['value'], h['value'], h['call_ms'], h['same_bytes'])"
done; done
