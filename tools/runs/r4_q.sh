mkdir -p gpurun_out/r4q
timeout -k 10 1100 python3 -m pytest tests -m gpu -x -q > gpurun_out/r4q/gpu_tests.log 2>&1
tail -6 gpurun_out/r4q/gpu_tests.log
echo done
