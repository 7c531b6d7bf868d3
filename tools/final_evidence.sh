#!/bin/bash
# final evidence of a round (GPU box, from the repo root): tools/final_evidence.sh TAG   (default r05)
TAG=${1:-r05}
set -x
timeout 2000 python -m pytest tests/ -m gpu -x -q > gpurun_out/${TAG}_gpu_suite.txt 2>&1; tail -3 gpurun_out/${TAG}_gpu_suite.txt
timeout 900 python tools/make_profiles.py $TAG > gpurun_out/${TAG}_make_profiles.log 2>&1; tail -25 gpurun_out/${TAG}_make_profiles.log
timeout 600 tools/bench_configs.sh > gpurun_out/${TAG}_configs.txt 2>&1; cat gpurun_out/${TAG}_configs.txt; cp gpurun_out/bench_configs.jsonl gpurun_out/${TAG}_configs.jsonl
timeout 600 python bench.py > gpurun_out/${TAG}_bench_default_run.json 2> gpurun_out/${TAG}_bench.err; cat gpurun_out/${TAG}_bench_default_run.json
timeout 300 python tools/latency.py > gpurun_out/${TAG}_latency.txt 2>&1; cat gpurun_out/${TAG}_latency.txt
make -C deflate-rs_amd -s stats
timeout 300 python tools/kernel_stats.py match > gpurun_out/${TAG}_k_match3_clock_shares.txt 2>&1; cat gpurun_out/${TAG}_k_match3_clock_shares.txt
timeout 300 python tools/kernel_stats.py sort > gpurun_out/${TAG}_k_sort_clock_shares.txt 2>&1; cat gpurun_out/${TAG}_k_sort_clock_shares.txt
# the multi-GPU call on this one device (dry run: the ranks' kernels queue on one GPU; what it shows is the host side of the exchanges)
for n in 2 4 8; do timeout 300 python bench.py --gpus $n --single-process --virtual --steps 5 --warmup 2 --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/${TAG}_multi_virtual_N$n.json; python -c "
import json; d=json.load(open('gpurun_out/${TAG}_multi_virtual_N$n.json')); print($n, d['value'], d['ms_per_step'], d.get('multi_trace_ms'))"; done
# BASELINE config 5 at 8 GiB: one GPU's range walk and eight ranks on this one device, against the oracle's digest
timeout 900 python tools/config5_8gib.py --virtual 8 2>/dev/null | tail -1 > gpurun_out/${TAG}_config5_8gib.json; cat gpurun_out/${TAG}_config5_8gib.json
# a timed region of more than a second (the default line's 20 steps are 0.1 s)
timeout 600 python bench.py --steps 300 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/${TAG}_bench_300_steps.json; python -c "
import json; d=json.load(open('gpurun_out/${TAG}_bench_300_steps.json')); print('300 steps:', d['value'], d['ms_per_step'], d['step_ms_events'], d['value_host_api'])"
