#!/bin/bash
# development aid (GPU box): the DESIGN section 5 table -- every BASELINE configuration that fits one GPU.
# Prints one summary line per configuration; the full bench lines go to gpurun_out/bench_configs.jsonl.
mkdir -p gpurun_out; : > gpurun_out/bench_configs.jsonl
run() { python bench.py --steps 5 --warmup 2 --no-live-pmc "$@" 2>/dev/null | tee -a gpurun_out/bench_configs.jsonl | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('%-44s %9.0f MB/s %7.2f ms  %s exact=%s cpu=%s' % (' '.join(sys.argv[1:]), d['value'], d['ms_per_step'], d['stage_ms'], d.get('bit_exact_vs_oracle'), (d.get('cpu_baseline') or {}).get('value')))" "$@"; }
run --workload zeros
run --workload zeros --level default
run --workload random
run --workload enwik8
run --workload enwik8 --level fast
run --workload enwik8 --level best
run --workload silesia
