#!/bin/bash
# usage: tools/pmc_run.sh TAG "COUNTERS..." [kernel-regex] [bench args...]   (run on the GPU box from the repo root)
TAG=$1; CTRS=$2; RX=${3:-k_match}; shift 3
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc $CTRS --kernel-include-regex "$RX" --output-format csv -d $R/gpurun_out/pmc_$TAG -- python $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline "$@" > $R/gpurun_out/pmc_$TAG.log 2>&1
python $R/tools/pmc_summary.py $R/gpurun_out/pmc_$TAG > $R/gpurun_out/pmc_$TAG.json
rm -rf $R/gpurun_out/pmc_$TAG
