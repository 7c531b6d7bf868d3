set -x
mkdir -p gpurun_out/r2
timeout 600 python tools/probes/pageable_call.py > gpurun_out/r2/pageable.txt 2>&1
cat gpurun_out/r2/pageable.txt
MI355_BOUNCE_TRACE=1 timeout 600 python tools/probes/pageable_call.py --threads 8 --reps 6 > gpurun_out/r2/pageable_trace.txt 2>&1
tail -12 gpurun_out/r2/pageable_trace.txt
timeout 900 python bench.py --gpus 2 --virtual --steps 5 --warmup 1 > gpurun_out/r2/bench_v2.json 2> gpurun_out/r2/bench_v2.err
tail -c 2500 gpurun_out/r2/bench_v2.json; tail -5 gpurun_out/r2/bench_v2.err
timeout 900 python bench.py --gpus 2 --virtual --single-process --steps 5 --warmup 1 > gpurun_out/r2/bench_sp2.json 2> gpurun_out/r2/bench_sp2.err
tail -c 2500 gpurun_out/r2/bench_sp2.json; tail -5 gpurun_out/r2/bench_sp2.err
