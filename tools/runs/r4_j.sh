mkdir -p gpurun_out/r4j
rm -f gpurun_out/r4j/sweep.txt
timeout -k 10 300 python3 -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r4j/smoke.log 2>&1 || { tail -20 gpurun_out/r4j/smoke.log; exit 1; }
tail -1 gpurun_out/r4j/smoke.log
for cfg in "VK_SIDE_START=event" "VK_SIDE_START=gate"; do
  echo "== $cfg" >> gpurun_out/r4j/sweep.txt
  env $cfg VK_SWEEP=10 timeout -k 10 300 python3 tools/ddp_slow_trace.py > gpurun_out/r4j/last.log 2>&1
  grep -E "no wrapper|wrapper instances|Error|error|Traceback" gpurun_out/r4j/last.log >> gpurun_out/r4j/sweep.txt
done
cat gpurun_out/r4j/sweep.txt
bash tools/ab_bench.sh "VK_SIDE_START=event VK_SIDE_START=gate" && cp gpurun_out/ab.txt gpurun_out/r4j/ab_gate.txt
timeout -k 10 900 python3 -m pytest tests/test_engine_gpu.py tests/test_training_gpu.py tests/test_fullsize_gpu.py tests/test_ddp_gpu.py -x -q > gpurun_out/r4j/tests.log 2>&1
tail -5 gpurun_out/r4j/tests.log
echo done
