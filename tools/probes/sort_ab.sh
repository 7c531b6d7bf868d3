#!/bin/bash
# development aid (GPU box): k_sort's average time on the bench's input for the product and every variants/v_*.so
R=$GRAFT_REPO_ROOT
for f in $R/deflate-rs_amd/libmi355deflate.so $R/deflate-rs_amd/variants/v_*.so; do
MI355_DEFLATE_LIB=$f timeout -s KILL 300 bash $R/tools/kstats.sh ab --no-live-pmc 2>&1 | grep -E "k_sort" | sed "s/^/$(basename $f)  /"
done
