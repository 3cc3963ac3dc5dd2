mkdir -p gpurun_out/r4u
timeout -k 10 600 python3 -m pytest tests/test_attention_gpu.py tests/test_engine_gpu.py tests/test_ctrl_golden_gpu.py tests/test_fp8_gpu.py -q > gpurun_out/r4u/tests.log 2>&1; tail -4 gpurun_out/r4u/tests.log
python3 bench.py --steps 12 --warmup 4 --no-cpu-baseline --no-kernel-timing --config ctrl_vl-bert_base --regions 100 > gpurun_out/r4u/vlbert_r100.json 2> gpurun_out/r4u/vlbert_r100.err; grep "timed region" gpurun_out/r4u/vlbert_r100.err
python3 bench.py --steps 12 --warmup 4 --no-cpu-baseline --no-kernel-timing --config ctrl_vl-bert_base --regions 100 --dtype fp8 > gpurun_out/r4u/vlbert_r100_fp8.json 2> gpurun_out/r4u/vlbert_r100_fp8.err; grep "timed region" gpurun_out/r4u/vlbert_r100_fp8.err
R=$(pwd)
( cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace -d /tmp/p_vl -o t -- python3 $R/bench.py --no-cpu-baseline --no-kernel-timing --config ctrl_vl-bert_base --regions 100 --steps 4 --warmup 2 --serial > $R/gpurun_out/r4u/vl.log 2>&1 && python3 $R/tools/rocprof_summary.py /tmp/p_vl/t_results.db $R/gpurun_out/r4u/vlbert_r100_kernel_stats_after )
head -12 gpurun_out/r4u/vlbert_r100_kernel_stats_after.md
echo done
