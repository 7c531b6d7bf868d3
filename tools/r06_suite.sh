# the whole GPU suite, the bench line, the N > 1 dry runs (one-GPU box)
mkdir -p gpurun_out/suite
timeout 1500 python -m pytest tests -x -q -m gpu > gpurun_out/suite/pytest_gpu.log 2>&1
tail -4 gpurun_out/suite/pytest_gpu.log
timeout 600 python bench.py > gpurun_out/suite/bench.json 2> gpurun_out/suite/bench.err
python - <<'PY'
import json
r=json.loads(open("gpurun_out/suite/bench.json").read().strip().splitlines()[-1])
print("value", r["value"], "ms", r["ms_per_step"], "host_api", r["value_host_api"]["value"], "pageable", r["value_host_api_pageable"]["value"], r["value_host_api_pageable"]["call_ms"], "runtime copies", r["value_host_api_pageable"]["runtime_copies"]["value"], "exact", r["bit_exact_vs_oracle"])
PY
timeout 900 python bench.py --gpus 2 --virtual --steps 5 --warmup 1 > gpurun_out/suite/bench_v2.json 2> gpurun_out/suite/bench_v2.err
tail -c 1200 gpurun_out/suite/bench_v2.json; tail -3 gpurun_out/suite/bench_v2.err
timeout 900 python bench.py --gpus 2 --virtual --single-process --steps 5 --warmup 1 > gpurun_out/suite/bench_sp2.json 2> gpurun_out/suite/bench_sp2.err
tail -c 1200 gpurun_out/suite/bench_sp2.json; tail -3 gpurun_out/suite/bench_sp2.err
