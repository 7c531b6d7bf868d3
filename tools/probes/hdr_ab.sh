#!/bin/bash
# development aid (GPU box): k_block_header of a 167 KB file for the product and every deflate-rs_amd/variants/v_*.so (kernel
# trace), the phase timers of the t_*.so builds (-DMI355_HDR_TIMERS), and the small-input latency table for each
R=$GRAFT_REPO_ROOT; cd /tmp && export TMPDIR=/tmp
for f in $R/deflate-rs_amd/libmi355deflate.so $R/deflate-rs_amd/variants/v_*.so; do
echo "== $(basename $f)"
MI355_DEFLATE_LIB=$f timeout -s KILL 200 bash $R/tools/probes/small_trace.sh 2>&1 | grep -E "k_block_header|k_small_tail|kernels "
MI355_DEFLATE_LIB=$f timeout -s KILL 200 python $R/tools/latency.py 2>&1 | grep -E "pg11|text 2 MB"
done
for f in $R/deflate-rs_amd/variants/t_*.so; do
echo "== $(basename $f)"
MI355_DEFLATE_LIB=$f timeout -s KILL 100 python $R/tools/probes/small_trace.py 3 2>&1 | grep "hdr timers" | tail -2
done
