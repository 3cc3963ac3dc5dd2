mkdir -p gpurun_out/r4g
for q in 4 8; do
  GPU_MAX_HW_QUEUES=$q timeout -k 10 400 python3 tools/ddp_overhead.py > gpurun_out/r4g/ddp_overhead_q$q.txt 2>&1
  grep -v "Gloo\|socket\|amdgpu.ids" gpurun_out/r4g/ddp_overhead_q$q.txt
done
bash tools/ab_bench.sh "GPU_MAX_HW_QUEUES=4 GPU_MAX_HW_QUEUES=8" && cp gpurun_out/ab.txt gpurun_out/r4g/ab_queues.txt
echo done
