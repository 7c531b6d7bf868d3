mkdir -p gpurun_out/r9
MI355_HOST_TRACE=1 MI355_BOUNCE_TRACE=1 timeout 600 python tools/probes/pageable_call.py --threads 8 --reps 3 > gpurun_out/r9/t1.txt 2>&1
grep -v amdgpu.ids gpurun_out/r9/t1.txt | tail -12
timeout 600 python tools/probes/pageable_call.py --threads 2,4,8,12 --reps 20 2>&1 | grep -v amdgpu.ids
MI355_BOUNCE_LAST_SUB=8 timeout 600 python tools/probes/pageable_call.py --threads 8 --reps 20 2>&1 | grep -v amdgpu.ids | tail -1
MI355_BOUNCE_LAST_SUB=1 timeout 600 python tools/probes/pageable_call.py --threads 8 --reps 20 2>&1 | grep -v amdgpu.ids | tail -1
