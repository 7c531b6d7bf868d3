mkdir -p gpurun_out/r10
MI355_HOST_TRACE=1 MI355_BOUNCE_TRACE=1 timeout 600 python tools/probes/pageable_call.py --threads 8 --reps 3 > gpurun_out/r10/t1.txt 2>&1
grep -v amdgpu.ids gpurun_out/r10/t1.txt | grep -v "transfers" | tail -9
timeout 600 python tools/probes/pageable_call.py --threads 4,8 --reps 20 2>&1 | grep -v amdgpu.ids
R=$PWD
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace -d $R/gpurun_out/r10/prof -o pg -- python $R/tools/probes/pageable_call.py --threads 8 --reps 5 > $R/gpurun_out/r10/prof.log 2>&1
cd $R && python tools/probes/chain.py gpurun_out/r10/prof/pg_results.db | tail -8
rm -f gpurun_out/r10/prof/pg_results.db
