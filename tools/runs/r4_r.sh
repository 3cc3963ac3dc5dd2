# round 4, final collection: kernel stats (serial / concurrent), PMC traffic, per-op table, timeline, bench line with cpu baseline
mkdir -p gpurun_out/r4r
bash tools/collect_profiles.sh > gpurun_out/r4r/collect.log 2>&1 || { tail -20 gpurun_out/r4r/collect.log; exit 1; }
tail -4 gpurun_out/r4r/collect.log
python3 bench.py --steps 6 --warmup 3 --no-cpu-baseline --serial --dump-ops gpurun_out/r4r/ops_per_launch.txt > gpurun_out/r4r/ops.log 2>&1 || exit 1
R=$(pwd)
( cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace -d /tmp/p_tl -o t -- python3 $R/bench.py --no-cpu-baseline --no-kernel-timing --steps 4 --warmup 2 > $R/gpurun_out/r4r/tl.log 2>&1 && python3 $R/tools/timeline.py /tmp/p_tl/t_results.db $R/gpurun_out/r4r/tl )
python3 bench.py --steps 20 --warmup 5 > gpurun_out/r4r/bench_final.json 2> gpurun_out/r4r/bench_final.err || exit 1
grep "timed region" gpurun_out/r4r/bench_final.err
echo done
