#!/bin/bash
# round-3 final evidence (GPU box, from the repo root)
set -x
timeout 2400 python -m pytest tests/ -m gpu -x -q > gpurun_out/r03f_gpu_suite.txt 2>&1; tail -3 gpurun_out/r03f_gpu_suite.txt
python tools/make_profiles.py r03_final > gpurun_out/r03f_make_profiles.log 2>&1; tail -25 gpurun_out/r03f_make_profiles.log
tools/bench_configs.sh > gpurun_out/r03_final_configs.txt 2>&1; cat gpurun_out/r03_final_configs.txt; cp gpurun_out/bench_configs.jsonl gpurun_out/r03_final_configs.jsonl
python bench.py > gpurun_out/r03_final_bench_default_run.json 2> gpurun_out/r03f_bench.err; cat gpurun_out/r03_final_bench_default_run.json
python tools/latency.py > gpurun_out/r03_final_latency.txt 2>&1; cat gpurun_out/r03_final_latency.txt
python tools/match3_stats.py > gpurun_out/r03_final_k_match3_clock_shares.txt 2>&1; cat gpurun_out/r03_final_k_match3_clock_shares.txt
