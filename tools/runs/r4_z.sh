mkdir -p gpurun_out/r4z
timeout -k 10 1100 python3 -m pytest tests -m gpu -x -q > gpurun_out/r4z/tests_full.log 2>&1
rc=$?; tail -4 gpurun_out/r4z/tests_full.log
if [ $rc -ne 0 ]; then exit $rc; fi
timeout -k 10 300 python3 -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -3
