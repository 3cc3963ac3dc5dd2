mkdir -p gpurun_out/r4v
R=$(pwd)
cd /tmp && export TMPDIR=/tmp
for cfg in "ctrl_lxmert 256" "ctrl_uniter_base 512" "ctrl_visualbert_base 256"; do
  set -- $cfg
  rocprofv3 --kernel-trace -d /tmp/p_$1 -o t -- python3 $R/bench.py --no-cpu-baseline --no-kernel-timing --config $1 --batch $2 --steps 4 --warmup 2 --serial > $R/gpurun_out/r4v/$1.log 2>&1
  python3 $R/tools/rocprof_summary.py /tmp/p_$1/t_results.db $R/gpurun_out/r4v/$1_kernel_stats
  grep "timed region" $R/gpurun_out/r4v/$1.log
  head -14 $R/gpurun_out/r4v/$1_kernel_stats.md
  rm -rf /tmp/p_$1
done
echo done
