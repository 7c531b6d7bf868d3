#!/bin/bash
# development aid (GPU box): the small-input latency table (pg11.txt, 2 MB) for the product and every deflate-rs_amd/variants/v_*.so,
# three times round so that a box's drift shows
R=$GRAFT_REPO_ROOT
for round in 1 2 3; do
for f in $R/deflate-rs_amd/libmi355deflate.so $R/deflate-rs_amd/variants/v_*.so; do
MI355_DEFLATE_LIB=$f timeout -s KILL 200 python $R/tools/latency.py 2>&1 | grep -E "pg11 167 KB        Default|text 2 MB          Default|random" | grep Default | sed "s/^/$(basename $f | cut -c1-16)  /; s/stages.*//"
done
done
