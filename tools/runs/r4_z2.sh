mkdir -p gpurun_out/r4z
timeout -k 10 600 python3 -m pytest tests/test_ddp_gpu.py tests/test_optim_gpu.py -x -q > gpurun_out/r4z/tests2.log 2>&1
tail -12 gpurun_out/r4z/tests2.log | cut -c1-220
