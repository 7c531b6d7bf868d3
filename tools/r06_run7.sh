mkdir -p gpurun_out/r7
MI355_BOUNCE_TRACE=1 MI355_BOUNCE_TRACE_D2H=1 timeout 600 python tools/probes/pageable_call.py --threads 8 --reps 2 > gpurun_out/r7/t1.txt 2>&1
grep -v amdgpu.ids gpurun_out/r7/t1.txt | tail -12
MI355_BOUNCE_TRACE=1 MI355_BOUNCE_SUB=32 timeout 600 python tools/probes/pageable_call.py --threads 8 --reps 3 > gpurun_out/r7/t2.txt 2>&1
grep -v amdgpu.ids gpurun_out/r7/t2.txt | tail -7
