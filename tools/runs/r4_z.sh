mkdir -p gpurun_out/r4z
timeout -k 10 1100 python3 -m pytest tests -m gpu -x -q > gpurun_out/r4z/tests_full.log 2>&1
rc=$?; tail -6 gpurun_out/r4z/tests_full.log
exit $rc
