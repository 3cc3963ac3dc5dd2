mkdir -p gpurun_out/r4z
timeout -k 10 900 python3 -m pytest tests/test_attention_gpu.py -x -q > gpurun_out/r4z/tests.log 2>&1
rc=$?; tail -3 gpurun_out/r4z/tests.log
if [ $rc -ne 0 ]; then exit 1; fi
VK_LIB=study timeout -k 10 300 python3 tools/bench_small.py fwdp > gpurun_out/r4z/attn.txt 2>&1
cat gpurun_out/r4z/attn.txt
