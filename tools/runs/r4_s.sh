# round 4, final: the whole GPU suite, then the profile set and the bench line of the committed sources
mkdir -p gpurun_out/r4s
timeout -k 10 900 python3 -m pytest tests -m gpu -q > gpurun_out/r4s/gpu_tests.log 2>&1
tail -4 gpurun_out/r4s/gpu_tests.log
bash tools/collect_profiles.sh > gpurun_out/r4s/collect.log 2>&1 || { tail -20 gpurun_out/r4s/collect.log; exit 1; }
tail -3 gpurun_out/r4s/collect.log
cp gpurun_out/prof_final/pmc_summary.json profiles/r04_pmc_summary.json
python3 bench.py --steps 6 --warmup 3 --no-cpu-baseline --serial --dump-ops gpurun_out/r4s/ops_per_launch.txt > gpurun_out/r4s/ops.log 2>&1 || exit 1
R=$(pwd)
( cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace -d /tmp/p_tl -o t -- python3 $R/bench.py --no-cpu-baseline --no-kernel-timing --steps 4 --warmup 2 > $R/gpurun_out/r4s/tl.log 2>&1 && python3 $R/tools/timeline.py /tmp/p_tl/t_results.db $R/gpurun_out/r4s/tl )
python3 bench.py --steps 20 --warmup 5 > gpurun_out/r4s/bench_final.json 2> gpurun_out/r4s/bench_final.err || exit 1
grep "timed region" gpurun_out/r4s/bench_final.err
echo done
