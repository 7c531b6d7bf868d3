set -x
mkdir -p gpurun_out/r1
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "pageable or streamed_in_pieces or pinned_host_range or multi_gpu_encode_in_one_call or device_resident" > gpurun_out/r1/pytest_new.log 2>&1
tail -5 gpurun_out/r1/pytest_new.log
timeout 600 python bench.py > gpurun_out/r1/bench.json 2> gpurun_out/r1/bench.err
tail -c 3000 gpurun_out/r1/bench.json
timeout 900 python bench.py --gpus 2 --virtual --steps 5 --warmup 1 > gpurun_out/r1/bench_v2.json 2> gpurun_out/r1/bench_v2.err
tail -c 2500 gpurun_out/r1/bench_v2.json; tail -5 gpurun_out/r1/bench_v2.err
