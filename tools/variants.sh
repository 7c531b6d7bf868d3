#!/bin/bash
# development aid (GPU box): time the stages for each build variant under deflate-rs_amd/variants/ (not the stats builds)
for f in deflate-rs_amd/variants/v_*.so; do
  MI355_DEFLATE_LIB=$PWD/$f timeout 120 python bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-host-api "$@" 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('%-44s' % '$f', d['value'], d['stage_ms'])"
done
