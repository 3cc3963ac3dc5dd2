mkdir -p gpurun_out/r4m
for q in 4 8 16; do GPU_MAX_HW_QUEUES=$q timeout -k 10 200 python3 tools/probe_streams.py 20 > gpurun_out/r4m/probe_q$q.txt 2>&1; grep -v amdgpu.ids gpurun_out/r4m/probe_q$q.txt; done
echo done
