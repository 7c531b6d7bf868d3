#!/bin/bash
# final evidence of a round (GPU box, from the repo root): tools/final_evidence.sh TAG   (default r04)
TAG=${1:-r04}
set -x
timeout 2000 python -m pytest tests/ -m gpu -x -q > gpurun_out/${TAG}_gpu_suite.txt 2>&1; tail -3 gpurun_out/${TAG}_gpu_suite.txt
timeout 900 python tools/make_profiles.py $TAG > gpurun_out/${TAG}_make_profiles.log 2>&1; tail -25 gpurun_out/${TAG}_make_profiles.log
timeout 600 tools/bench_configs.sh > gpurun_out/${TAG}_configs.txt 2>&1; cat gpurun_out/${TAG}_configs.txt; cp gpurun_out/bench_configs.jsonl gpurun_out/${TAG}_configs.jsonl
timeout 600 python bench.py > gpurun_out/${TAG}_bench_default_run.json 2> gpurun_out/${TAG}_bench.err; cat gpurun_out/${TAG}_bench_default_run.json
timeout 300 python tools/latency.py > gpurun_out/${TAG}_latency.txt 2>&1; cat gpurun_out/${TAG}_latency.txt
timeout 300 python tools/kernel_stats.py match > gpurun_out/${TAG}_k_match3_clock_shares.txt 2>&1; cat gpurun_out/${TAG}_k_match3_clock_shares.txt
timeout 300 python tools/kernel_stats.py sort > gpurun_out/${TAG}_k_sort_clock_shares.txt 2>&1; cat gpurun_out/${TAG}_k_sort_clock_shares.txt
# the multi-GPU call on this one device (dry run: the ranks' kernels queue on one GPU; what it shows is the host side of the exchanges)
for n in 2 4 8; do timeout 300 python bench.py --gpus $n --single-process --virtual --steps 5 --warmup 2 --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/${TAG}_multi_virtual_N$n.json; python -c "
import json; d=json.load(open('gpurun_out/${TAG}_multi_virtual_N$n.json')); print($n, d['value'], d['ms_per_step'], d.get('multi_trace_ms'))"; done
