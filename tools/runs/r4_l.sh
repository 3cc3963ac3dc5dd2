mkdir -p gpurun_out/r4l
rm -f gpurun_out/r4l/sweep.txt
for cfg in "VK_SIDE_START=event VK_DDP_ORDER=event" "VK_SIDE_START=event VK_DDP_ORDER=gate" "VK_SIDE_START=gate VK_DDP_ORDER=gate"; do
  echo "== $cfg" >> gpurun_out/r4l/sweep.txt
  env $cfg VK_SWEEP=8 timeout -k 10 300 python3 tools/ddp_slow_trace.py > gpurun_out/r4l/last.log 2>&1
  grep -E "no wrapper|wrapper instances|Error|error|Traceback" gpurun_out/r4l/last.log >> gpurun_out/r4l/sweep.txt
done
cat gpurun_out/r4l/sweep.txt
echo done
