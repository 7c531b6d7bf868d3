mkdir -p gpurun_out/r6
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O2 -mclflushopt -o /tmp/d2h_ring tools/probes/d2h_ring.hip -lhsa-runtime64 -lpthread 2>&1 | grep -v warning | head -5
timeout 300 /tmp/d2h_ring > gpurun_out/r6/d2h_ring.txt 2>&1
cat gpurun_out/r6/d2h_ring.txt
