# round 4, call A: any-order launch probe, write-through GEMM stores A/B, per-op tables
mkdir -p gpurun_out
./tools/micro/anyorder > gpurun_out/anyorder.txt 2>&1
cat gpurun_out/anyorder.txt
bash tools/ab_bench.sh "VK_LIB= VK_LIB=wt VK_LIB=wt2" || exit 1
python3 bench.py --steps 6 --warmup 3 --no-cpu-baseline --serial --dump-ops gpurun_out/ops_base.txt > gpurun_out/ops_base.log 2>&1 || exit 1
VK_LIB=wt python3 bench.py --steps 6 --warmup 3 --no-cpu-baseline --serial --dump-ops gpurun_out/ops_wt.txt > gpurun_out/ops_wt.log 2>&1 || exit 1
echo done
