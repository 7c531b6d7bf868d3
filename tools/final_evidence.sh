#!/bin/bash
# final evidence of a round (GPU box, from the repo root): tools/final_evidence.sh TAG   (default r06)
TAG=${1:-r06}
mkdir -p gpurun_out
set -x
timeout -s KILL 2000 python -m pytest tests/ -m gpu -x -q > gpurun_out/${TAG}_gpu_suite.txt 2>&1; tail -3 gpurun_out/${TAG}_gpu_suite.txt
timeout -s KILL 900 python tools/make_profiles.py $TAG > gpurun_out/${TAG}_make_profiles.log 2>&1; tail -25 gpurun_out/${TAG}_make_profiles.log
timeout -s KILL 600 tools/bench_configs.sh > gpurun_out/${TAG}_configs.txt 2>&1; cat gpurun_out/${TAG}_configs.txt; cp gpurun_out/bench_configs.jsonl gpurun_out/${TAG}_configs.jsonl
timeout -s KILL 600 python bench.py > gpurun_out/${TAG}_bench_default_run.json 2> gpurun_out/${TAG}_bench.err; cat gpurun_out/${TAG}_bench_default_run.json
timeout -s KILL 300 python tools/latency.py > gpurun_out/${TAG}_latency.txt 2>&1; cat gpurun_out/${TAG}_latency.txt
# the kernels of one 167 KB call: start, duration, gap to the one before (nine launches, back to back)
timeout -s KILL 200 bash tools/probes/small_trace.sh > gpurun_out/${TAG}_small_call_trace.txt 2>&1; cat gpurun_out/${TAG}_small_call_trace.txt
timeout -s KILL 300 python tools/config4_segments.py 2>/dev/null > gpurun_out/${TAG}_config4_segments.txt; cat gpurun_out/${TAG}_config4_segments.txt
timeout -s KILL 300 python tools/probes/pageable_call.py --threads 4,8 --reps 20 2>&1 | grep -v amdgpu.ids > gpurun_out/${TAG}_pageable_call.txt; cat gpurun_out/${TAG}_pageable_call.txt
timeout -s KILL 200 bash tools/probes/lds_conflicts.sh records96 records256 records40 text dbrows > gpurun_out/${TAG}_lds_conflicts.txt 2>&1; cat gpurun_out/${TAG}_lds_conflicts.txt
make -C deflate-rs_amd -s stats
timeout -s KILL 300 python tools/kernel_stats.py match > gpurun_out/${TAG}_k_match3_clock_shares.txt 2>&1; cat gpurun_out/${TAG}_k_match3_clock_shares.txt
timeout -s KILL 300 python tools/kernel_stats.py sort > gpurun_out/${TAG}_k_sort_clock_shares.txt 2>&1; cat gpurun_out/${TAG}_k_sort_clock_shares.txt
# N > 1 on this one device (dry runs: the ranks' kernels queue on one GPU; what they show is that the driver's command runs, the
# exchanges' host side, and the digests): the driver's form (one process per rank; gloo here), and the one-call form
for n in 2 4 8; do timeout -s KILL 900 python bench.py --gpus $n --virtual --steps 3 --warmup 1 2>/dev/null | tail -1 > gpurun_out/${TAG}_multi_procs_virtual_N$n.json; python -c "
import json; d=json.load(open('gpurun_out/${TAG}_multi_procs_virtual_N$n.json')); print('procs', $n, d['value'], d['ms_per_step'], d.get('stitch'), d.get('rccl_ranks'), d.get('bit_exact_vs_oracle'), d.get('oracle_digest'))"; done
for n in 2 4 8; do timeout -s KILL 900 python bench.py --gpus $n --single-process --virtual --steps 3 --warmup 1 2>/dev/null | tail -1 > gpurun_out/${TAG}_multi_virtual_N$n.json; python -c "
import json; d=json.load(open('gpurun_out/${TAG}_multi_virtual_N$n.json')); print('one call', $n, d['value'], d['ms_per_step'], d.get('stitch'), d.get('rccl_ranks'), d.get('bit_exact_vs_oracle'), d.get('multi_phases_ms_rank0'))"; done
# BASELINE config 5 at 8 GiB: one GPU's range walk and eight ranks on this one device, against the oracle's digest
timeout -s KILL 900 python tools/config5_8gib.py --virtual 8 2>/dev/null | tail -1 > gpurun_out/${TAG}_config5_8gib.json; cat gpurun_out/${TAG}_config5_8gib.json
# a timed region of more than a second (the default line's 20 steps are 0.1 s)
timeout -s KILL 600 python bench.py --steps 300 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/${TAG}_bench_300_steps.json; python -c "
import json; d=json.load(open('gpurun_out/${TAG}_bench_300_steps.json')); print('300 steps:', d['value'], d['ms_per_step'], d['step_ms_events'], d['value_host_api'], d['value_host_api_pageable'])"
# the fuzzers, smoke() and the multi-GPU self-check
set +x
(echo "# the fuzzers, smoke() and tools/multi_selfcheck.py on the round's final tree (MI355X box)"
 timeout -s KILL 900 python tools/fuzz_gpu.py 2000 60000 2>&1 | tail -1
 timeout -s KILL 600 python tools/fuzz_flush_gaps.py 300 1 2>&1 | tail -1
 timeout -s KILL 600 python tools/fuzz_shard.py 60 2>&1 | tail -1
 timeout -s KILL 500 python tools/fuzz_flushed_ranges.py 8 2>&1 | tail -1
 timeout -s KILL 100 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
 timeout -s KILL 200 python tools/multi_selfcheck.py 2>&1 | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('multi_selfcheck: ok=%s devices=%d cases=%d all same=%s (%.1f s)' % (d['ok'], d['devices'], len(d['cases']), all(c['same'] for c in d['cases']), d['seconds']))"
 timeout -s KILL 600 python tools/fuzz_pageable.py 2>&1 | tail -1) > gpurun_out/${TAG}_fuzz_round_end.txt 2>&1; cat gpurun_out/${TAG}_fuzz_round_end.txt
set -x
