mkdir -p gpurun_out/r12
for kb in 2048 4096 8192; do
echo "run_min $kb KiB"
MI355_BOUNCE_RUN_MIN_KB=$kb timeout 600 python tools/probes/pageable_call.py --threads 4,8 --reps 20 2>&1 | grep -v amdgpu.ids | tail -2
done
MI355_HOST_TRACE=1 MI355_BOUNCE_TRACE=1 timeout 600 python tools/probes/pageable_call.py --threads 8 --reps 3 > gpurun_out/r12/t1.txt 2>&1
grep -v amdgpu.ids gpurun_out/r12/t1.txt | grep "host threads" | tail -3
grep -v amdgpu.ids gpurun_out/r12/t1.txt | grep "transfers" | tail -1
