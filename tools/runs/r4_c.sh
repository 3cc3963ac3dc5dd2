# round 4, call C: soft-boundary FFN pair -- tests (bounded), then A/B in the step
mkdir -p gpurun_out/r4c
timeout -k 10 400 python3 -m pytest tests/test_gemm_gpu.py -x -q -k "soft_boundary" > gpurun_out/r4c/soft_tests.log 2>&1
rc=$?
tail -15 gpurun_out/r4c/soft_tests.log
if [ $rc -ne 0 ]; then echo "soft tests failed rc=$rc"; exit 1; fi
bash tools/ab_bench.sh "VK_SOFT=0 VK_SOFT=1" || exit 1
cp gpurun_out/ab.txt gpurun_out/r4c/ab.txt
VK_SOFT=1 python3 bench.py --steps 6 --warmup 3 --no-cpu-baseline --serial --dump-ops gpurun_out/r4c/ops_soft.txt > gpurun_out/r4c/ops_soft.log 2>&1 || exit 1
timeout -k 10 600 python3 -m pytest tests/test_gemm_gpu.py tests/test_engine_gpu.py -x -q > gpurun_out/r4c/gemm_engine_tests.log 2>&1
tail -5 gpurun_out/r4c/gemm_engine_tests.log
echo done
