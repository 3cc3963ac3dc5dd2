mkdir -p gpurun_out/r4k
VK_SIDE_START=event VK_SWEEP=10 timeout -k 10 300 python3 tools/ddp_slow_trace.py > gpurun_out/r4k/last.log 2>&1
grep -E "no wrapper|instance|Error|error|Traceback" gpurun_out/r4k/last.log
echo done
