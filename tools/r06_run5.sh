mkdir -p gpurun_out/r5
MI355_BOUNCE_TRACE=1 timeout 600 python tools/probes/pageable_call.py --threads 8 --reps 5 > gpurun_out/r5/pageable_trace_8.txt 2>&1
grep -v amdgpu.ids gpurun_out/r5/pageable_trace_8.txt | tail -7
MI355_D2H_ENGINE=-1 timeout 600 python tools/probes/pageable_call.py --threads 8 --reps 20 2>&1 | grep -v amdgpu.ids
timeout 600 python tools/probes/pageable_call.py --threads 2,4,8,12 --reps 20 2>&1 | grep -v amdgpu.ids
