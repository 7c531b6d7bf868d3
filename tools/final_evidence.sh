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
