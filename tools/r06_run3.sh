mkdir -p gpurun_out/r3
for t in 2 8; do
MI355_BOUNCE_TRACE=1 timeout 600 python tools/probes/pageable_call.py --threads $t --reps 8 > gpurun_out/r3/pageable_trace_$t.txt 2>&1
grep -v amdgpu.ids gpurun_out/r3/pageable_trace_$t.txt | tail -14
done
lscpu | grep -i "numa\|socket\|model name" 
