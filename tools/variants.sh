#!/bin/bash
# development aid: time the stages for each build variant under deflate-rs_amd/variants/
for f in deflate-rs_amd/variants/*.so; do
  MI355_DEFLATE_LIB=$PWD/$f python bench.py --steps 3 --warmup 1 --no-cpu-baseline "$@" 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$f', d['value'], d['stage_ms'])"
done
