#!/bin/bash
# development aid (GPU box): the C example many times, W workers in parallel, under the fault trap of
# tools/probes/segv_trap.c (backtrace of the faulting thread on stderr); the same number of runs of a HIP
# program that does not use the library (tools/probes/hip_control.hip) as the control.
#   tools/cli_crash.sh [runs per worker = 250] [workers = 4] [control: 1/0]
N=${1:-250}; W=${2:-4}; CTL=${3:-1}
OUT=gpurun_out/cli_crash; mkdir -p $OUT; rm -f $OUT/*
gcc -g -O2 -std=c99 -Iinclude examples/mi355_deflate_cli.c -Ldeflate-rs_amd -lmi355deflate -Wl,-rpath,$PWD/deflate-rs_amd -o /tmp/cli || exit 1
gcc -O1 -g -shared -fPIC -o /tmp/segv_trap.so tools/probes/segv_trap.c -ldl || exit 1
[ "$CTL" = 1 ] && { /opt/rocm/bin/hipcc -w --offload-arch=gfx950 -O2 -o /tmp/hip_control tools/probes/hip_control.hip || exit 1; }
ulimit -c 0
IN=tests/golden/ref_inputs/pg11.txt
worker() {
  local w=$1 fails=0
  for i in $(seq 1 $N); do
    case $((i % 4)) in
      0) args="-zlib -best -chunk 5000";; 1) args="-zlib -best -chunk 5000";; 2) args="-raw -default";; 3) args="-gzip -fast -chunk 5000";;
    esac
    LD_PRELOAD=/tmp/segv_trap.so /tmp/cli $args $IN /tmp/out.$w.bin 2>/tmp/err.$w.txt; rc=$?
    if [ $rc -ne 0 ]; then fails=$((fails+1)); { echo "=== worker $w run $i [$args] rc=$rc"; cat /tmp/err.$w.txt; } >> $OUT/cli_fail.$w.txt; fi
  done
  echo "cli worker $w: fails=$fails of $N" >> $OUT/summary.txt
}
control() {
  local w=$1 fails=0
  for i in $(seq 1 $N); do
    LD_PRELOAD=/tmp/segv_trap.so /tmp/hip_control 2>/tmp/cerr.$w.txt; rc=$?
    if [ $rc -ne 0 ]; then fails=$((fails+1)); { echo "=== control $w run $i rc=$rc"; cat /tmp/cerr.$w.txt; } >> $OUT/ctl_fail.$w.txt; fi
  done
  echo "control worker $w: fails=$fails of $N" >> $OUT/summary.txt
}
t0=$(date +%s)
for w in $(seq 1 $W); do worker $w & done; wait
t1=$(date +%s); echo "cli: $((W*N)) runs in $((t1-t0)) s" >> $OUT/summary.txt
if [ "$CTL" = 1 ]; then for w in $(seq 1 $W); do control $w & done; wait; t2=$(date +%s); echo "control: $((W*N)) runs in $((t2-t1)) s" >> $OUT/summary.txt; fi
cat $OUT/summary.txt; head -c 6000 $OUT/cli_fail.*.txt 2>/dev/null; head -c 3000 $OUT/ctl_fail.*.txt 2>/dev/null
