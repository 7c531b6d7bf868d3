#!/bin/bash
# development aid (GPU box): the C example, many runs
gcc -g -O2 -std=c99 -Iinclude examples/mi355_deflate_cli.c -Ldeflate-rs_amd -lmi355deflate -Wl,-rpath,$PWD/deflate-rs_amd -o /tmp/cli || exit 1
ulimit -c 0
fails=0
for i in $(seq 1 25); do
  for args in "-raw -default" "-raw -default -chunk 5000" "-zlib -best" "-zlib -best -chunk 5000" "-gzip -fast" "-gzip -fast -chunk 5000"; do
    /tmp/cli $args tests/golden/ref_inputs/pg11.txt /tmp/out.bin 2>/dev/null; rc=$?
    if [ $rc -ne 0 ]; then echo "run $i [$args] rc=$rc"; fails=$((fails+1)); fi
  done
done
echo "fails=$fails of 150"
if [ $fails -gt 0 ]; then
  for i in $(seq 1 30); do /opt/rocm/bin/rocgdb -batch -ex run -ex bt --args /tmp/cli -zlib -best -chunk 5000 tests/golden/ref_inputs/pg11.txt /tmp/out.bin 2>&1 | grep -A25 "SIGSEGV" | head -40 && break; done
fi
