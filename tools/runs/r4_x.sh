mkdir -p gpurun_out/r4x
: > gpurun_out/r4x/ab.txt
for rep in 1 2; do
  python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-kernel-timing 2>&1 | grep "timed region" | sed "s/^/dropout on : /" >> gpurun_out/r4x/ab.txt
  python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-kernel-timing --no-dropout 2>&1 | grep "timed region" | sed "s/^/dropout OFF: /" >> gpurun_out/r4x/ab.txt
done
cat gpurun_out/r4x/ab.txt
echo done
