mkdir -p gpurun_out/r4z
timeout -k 10 300 python3 tools/bench_adamw.py > gpurun_out/r4z/adamw.txt 2>&1
cat gpurun_out/r4z/adamw.txt
