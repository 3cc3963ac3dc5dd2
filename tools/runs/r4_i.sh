mkdir -p gpurun_out/r4i
rm -f gpurun_out/r4i/sweep.txt
for cfg in "VK_X=0" "VK_SIDE_DELAY_US=8" "VK_SIDE_DELAY_US=20" "VK_SIDE_PRIORITY=low"; do
  echo "== $cfg" >> gpurun_out/r4i/sweep.txt
  env $cfg VK_SWEEP=10 timeout -k 10 300 python3 tools/ddp_slow_trace.py > gpurun_out/r4i/last.log 2>&1
  grep -E "no wrapper|wrapper instances|Error|error|Traceback" gpurun_out/r4i/last.log >> gpurun_out/r4i/sweep.txt
  tail -3 gpurun_out/r4i/last.log >> gpurun_out/r4i/sweep.txt
done
cat gpurun_out/r4i/sweep.txt
echo done
