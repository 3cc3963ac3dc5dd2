mkdir -p gpurun_out/r4o
VK_SWEEP=10 timeout -k 10 300 python3 tools/ddp_slow_trace.py > gpurun_out/r4o/last.log 2>&1
grep -E "no wrapper|wrapper instances|Error|error|Traceback|Warning" gpurun_out/r4o/last.log
bash tools/ab_bench.sh "VK_STREAM_PROBE=0 VK_STREAM_PROBE=1" && cp gpurun_out/ab.txt gpurun_out/r4o/ab_probe.txt
timeout -k 10 400 python3 tools/comm_footprint.py --reserve 0,16,32 > gpurun_out/r4o/comm_footprint.txt 2>&1
grep -v "Gloo\|socket\|amdgpu" gpurun_out/r4o/comm_footprint.txt
echo done
