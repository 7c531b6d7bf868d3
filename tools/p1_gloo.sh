#!/bin/bash
# development aid (GPU box): the N-rank stream-exact step as a dry run on ONE GPU (gloo carries the exchanges through
# host memory, the ranks share the device): bench lines with p1_phases_ms_rank0 -> gpurun_out/<tag>_p1_gloo_N<k>.json
#   tools/p1_gloo.sh TAG [bench args]
TAG=${1:-r03}; shift
mkdir -p gpurun_out
for N in 2 4 8; do
  MI355_BENCH_BACKEND=gloo HSA_ENABLE_IPC_MODE_LEGACY=0 timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node $N \
    --master-addr 127.0.0.1 --master-port $((29500 + N)) bench.py --gpus $N --steps 5 --warmup 2 --no-cpu-baseline "$@" \
    2>gpurun_out/${TAG}_p1_gloo_N$N.err | tail -1 > gpurun_out/${TAG}_p1_gloo_N$N.json
  python - <<PY
import json
try:
    d = json.load(open("gpurun_out/${TAG}_p1_gloo_N$N.json"))
    print("N=$N", d["value"], "MB/s", d["ms_per_step"], "ms/step", d.get("p1_phases_ms_rank0"))
except Exception as e:
    print("N=$N failed:", e); print(open("gpurun_out/${TAG}_p1_gloo_N$N.err").read()[-2000:])
PY
done
