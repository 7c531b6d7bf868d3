mkdir -p gpurun_out/r8
MI355_HOST_TRACE=1 MI355_BOUNCE_SUB=32 timeout 600 python tools/probes/pageable_call.py --threads 8 --reps 3 > gpurun_out/r8/t1.txt 2>&1
grep -v amdgpu.ids gpurun_out/r8/t1.txt | grep -v "^\[bounce\]" | tail -14
