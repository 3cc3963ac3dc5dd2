# round 4, call E: vendor yardstick under the cold protocol (item 4), communication footprint rehearsal (item 5), bench line of the drop-in step
mkdir -p gpurun_out/r4e
python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-kernel-timing > gpurun_out/r4e/bench_dropin.json 2> gpurun_out/r4e/bench_dropin.err || exit 1
grep "timed region" gpurun_out/r4e/bench_dropin.err
timeout -k 10 500 python3 tools/bench_shapes.py 0 > gpurun_out/r4e/gemm_shapes.txt 2>&1
cat gpurun_out/r4e/gemm_shapes.txt
timeout -k 10 400 python3 tools/comm_footprint.py --reserve 0,16,32 > gpurun_out/r4e/comm_footprint.txt 2>&1
cat gpurun_out/r4e/comm_footprint.txt
echo done
