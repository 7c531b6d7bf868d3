mkdir -p gpurun_out/r4
MI355_BOUNCE_TRACE=1 timeout 600 python tools/probes/pageable_call.py --threads 8 --reps 5 > gpurun_out/r4/pageable_trace_8.txt 2>&1
grep -v amdgpu.ids gpurun_out/r4/pageable_trace_8.txt | tail -9
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --memory-copy-trace --stats -d /root/repo/gpurun_out/r4/prof -o pg -- python /root/repo/tools/probes/pageable_call.py --threads 8 --reps 5 > /root/repo/gpurun_out/r4/prof.log 2>&1
ls /root/repo/gpurun_out/r4/prof/* | head; 
