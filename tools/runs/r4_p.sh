mkdir -p gpurun_out/r4p
timeout -k 10 500 python3 -m pytest tests/test_gemm_gpu.py -x -q -k "chain or soft_boundary" > gpurun_out/r4p/chain_tests.log 2>&1
rc=$?
tail -8 gpurun_out/r4p/chain_tests.log
if [ $rc -ne 0 ]; then echo "chain tests failed rc=$rc"; exit 1; fi
VK_LIB=study timeout -k 10 200 python3 tools/stamp_soft.py chain > gpurun_out/r4p/chain_stamps.txt 2>&1
grep -v amdgpu.ids gpurun_out/r4p/chain_stamps.txt
VK_LIB=study timeout -k 10 200 python3 tools/stamp_soft.py > gpurun_out/r4p/pair_stamps.txt 2>&1
grep "fenced it 3\|soft   it 3" gpurun_out/r4p/pair_stamps.txt
bash tools/ab_bench.sh "VK_CHAIN=0 VK_CHAIN=fwd VK_CHAIN=all" && cp gpurun_out/ab.txt gpurun_out/r4p/ab_chain.txt
timeout -k 10 400 python3 -m pytest tests/test_engine_gpu.py -x -q > gpurun_out/r4p/engine_default.log 2>&1; tail -2 gpurun_out/r4p/engine_default.log
VK_CHAIN=all timeout -k 10 400 python3 -m pytest tests/test_engine_gpu.py tests/test_fullsize_gpu.py -x -q > gpurun_out/r4p/engine_chain.log 2>&1; tail -2 gpurun_out/r4p/engine_chain.log
echo done
