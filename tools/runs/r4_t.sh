mkdir -p gpurun_out/r4t
timeout -k 10 400 python3 -m pytest tests/test_ddp_gpu.py tests/test_streams_gpu.py -q > gpurun_out/r4t/tests.log 2>&1; tail -4 gpurun_out/r4t/tests.log
bash tools/ab_bench.sh "VK_CHAIN=0 VK_CHAIN=fwd" && cp gpurun_out/ab.txt gpurun_out/r4t/ab_chain.txt
echo done
