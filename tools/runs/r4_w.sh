# round 4, closing session: configuration sweep, then the profile set and bench line of the final sources
mkdir -p gpurun_out/r4w3
bash tools/bench_configs.sh > gpurun_out/r4w3/configs.txt 2>&1
cat gpurun_out/r4w3/configs.txt
bash tools/collect_profiles.sh > gpurun_out/r4w3/collect.log 2>&1 || { tail -20 gpurun_out/r4w3/collect.log; exit 1; }
tail -3 gpurun_out/r4w3/collect.log
cp gpurun_out/prof_final/pmc_summary.json profiles/r04_pmc_summary.json
python3 bench.py --steps 6 --warmup 3 --no-cpu-baseline --serial --dump-ops gpurun_out/r4w3/ops_per_launch.txt > gpurun_out/r4w3/ops.log 2>&1 || exit 1
R=$(pwd)
( cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace -d /tmp/p_tl -o t -- python3 $R/bench.py --no-cpu-baseline --no-kernel-timing --steps 4 --warmup 2 > $R/gpurun_out/r4w3/tl.log 2>&1 && python3 $R/tools/timeline.py /tmp/p_tl/t_results.db $R/gpurun_out/r4w3/tl )
python3 bench.py --steps 20 --warmup 5 > gpurun_out/r4w3/bench_final.json 2> gpurun_out/r4w3/bench_final.err || exit 1
grep "timed region" gpurun_out/r4w3/bench_final.err
echo done
timeout -k 10 400 python3 tools/comm_footprint.py --reserve 0 > gpurun_out/r4w3/comm_footprint.txt 2>&1 || { tail -5 gpurun_out/r4w3/comm_footprint.txt; exit 1; }
tail -4 gpurun_out/r4w3/comm_footprint.txt
