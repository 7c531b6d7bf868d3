#!/bin/bash
# development aid (GPU box): a fast parity subset, then the bench line's numbers
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "fixtures or known or edge or random_block or mixed or text_like or p1_sharded or zlib or flush or config3_full_size_equals or custom or periodic" 2>&1 | tail -4
for a in "$@" ""; do
python bench.py --steps 5 --warmup 2 --no-cpu-baseline $a 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print(sys.argv[1:], d['value'], d['ms_per_step'], d['stage_ms'])" "$a"
[ -z "$a" ] && break
done
