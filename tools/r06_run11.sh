mkdir -p gpurun_out/r11
for st in 1 2 3; do for kb in 1024 4096; do
echo "streams $st run_min $kb KiB"
MI355_BOUNCE_STREAMS=$st MI355_BOUNCE_RUN_MIN_KB=$kb timeout 600 python tools/probes/pageable_call.py --threads 8 --reps 20 2>&1 | grep -v amdgpu.ids | tail -1
done; done
echo "streams 2 run_min 256"
MI355_BOUNCE_STREAMS=2 MI355_BOUNCE_RUN_MIN_KB=256 timeout 600 python tools/probes/pageable_call.py --threads 8 --reps 20 2>&1 | grep -v amdgpu.ids | tail -1
R=$PWD
cd /tmp && export TMPDIR=/tmp
MI355_BOUNCE_STREAMS=2 timeout 600 rocprofv3 --kernel-trace -d $R/gpurun_out/r11/prof -o pg -- python $R/tools/probes/pageable_call.py --threads 8 --reps 5 > $R/gpurun_out/r11/prof.log 2>&1
cd $R && python tools/probes/chain.py gpurun_out/r11/prof/pg_results.db | tail -5
rm -f gpurun_out/r11/prof/pg_results.db
