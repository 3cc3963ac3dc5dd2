# A/B of environment switches on bench.py in ONE box session: tools/ab_bench.sh "VAR=a VAR=b ..." [extra bench args]
# every variant runs twice, interleaved; prints ms/step per run.  Variant `base` runs .base/bench.py: a second tree built beside this one
# (`git worktree add --detach .base <rev> && make -C .base/volta_amd/csrc`; git-ignored, travels to the GPU box), `serial` the serial schedule,
# VK_LIB=<name> a second build of the library (volta_amd/libvolta_hip_<name>.so).
R=$(pwd)
OUT=$R/gpurun_out/ab.txt
: > $OUT
for rep in 1 2; do
  for v in $1; do
    if [ "$v" = "base" ]; then
      python3 $R/.base/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-kernel-timing $2 2>&1 | grep "timed region" | sed "s/^/base (.base tree): /" >> $OUT
    elif [ "$v" = "serial" ]; then
      python3 $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-kernel-timing --serial $2 2>&1 | grep "timed region" | sed "s/^/serial: /" >> $OUT
    else
      env $v python3 $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-kernel-timing $2 2>&1 | grep "timed region" | sed "s/^/$v: /" >> $OUT
    fi
  done
done
cat $OUT
