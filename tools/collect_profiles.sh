# Collects the rocprofv3 summaries kept under profiles/ (run on the GPU box from the repo root; outputs under gpurun_out/prof_final/).
# One program per rocprofv3 call, kernel trace and PMC passes kept separate (no --pmc together with other trace domains).
set -e
R=$(pwd)
OUT=$R/gpurun_out/prof_final
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
B="$R/bench.py --no-cpu-baseline --no-kernel-timing"
rocprofv3 --kernel-trace -d /tmp/p_serial -o t -- python3 $B --steps 6 --warmup 3 --serial > $OUT/serial.log 2>&1
python3 $R/tools/rocprof_summary.py /tmp/p_serial/t_results.db $OUT/kernel_stats_serial
rm -rf /tmp/p_serial
echo "serial trace done" >> $OUT/progress.txt
rocprofv3 --kernel-trace -d /tmp/p_conc -o t -- python3 $B --steps 6 --warmup 3 > $OUT/concurrent.log 2>&1
python3 $R/tools/rocprof_summary.py /tmp/p_conc/t_results.db $OUT/kernel_stats_concurrent
rm -rf /tmp/p_conc
echo "concurrent trace done" >> $OUT/progress.txt
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $c -d /tmp/p_$c -o t -- python3 $B --steps 1 --warmup 1 --serial > $OUT/pmc_$c.log 2>&1
  python3 $R/tools/rocprof_summary.py /tmp/p_$c/t_results.db $OUT/pmc_$c --pmc $c
  rm -rf /tmp/p_$c
  echo "pmc $c done" >> $OUT/progress.txt
done
python3 $R/tools/pmc_traffic.py $OUT/pmc_FETCH_SIZE.csv $OUT/pmc_WRITE_SIZE.csv 2 $OUT/pmc_hbm_traffic.md $OUT/pmc_summary.json
grep -h "timed region" $OUT/*.log || true
