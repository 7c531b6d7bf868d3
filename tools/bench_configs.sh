#!/bin/bash
# development aid (GPU box): the DESIGN section 5 table -- every BASELINE configuration that fits one GPU
run() { python bench.py --steps 3 --warmup 1 "$@" 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('%-44s %9.0f MB/s %7.2f ms  %s exact=%s' % (' '.join(sys.argv[1:]), d['value'], d['ms_per_step'], d['stage_ms'], d.get('bit_exact_vs_oracle')))" "$@"; }
run --workload zeros
run --workload zeros --level default
run --workload random
run --workload enwik8
run --workload enwik8 --level fast
run --workload enwik8 --level best
run --workload silesia
