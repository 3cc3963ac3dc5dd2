mkdir -p gpurun_out/r4z
timeout -k 10 900 python3 -m pytest tests/test_gemm_gpu.py -x -q > gpurun_out/r4z/tests.log 2>&1
rc=$?; tail -3 gpurun_out/r4z/tests.log
if [ $rc -ne 0 ]; then exit 1; fi
for i in 1 2; do
for v in base ""; do
VK_LIB=$v timeout -k 10 300 python3 bench.py --no-cpu-baseline --no-kernel-timing > gpurun_out/r4z/b.json 2> gpurun_out/r4z/b.err || exit 1
python3 -c "
import json; d=json.loads(open('gpurun_out/r4z/b.json').read().strip().splitlines()[-1]); print('[lib $v]', round(d['ms_per_step'],3))"
done
done
for r in 8 16; do
VK_RESERVE_CUS=$r timeout -k 10 300 python3 bench.py --no-cpu-baseline --no-kernel-timing > gpurun_out/r4z/b.json 2> gpurun_out/r4z/b.err || exit 1
python3 -c "
import json; d=json.loads(open('gpurun_out/r4z/b.json').read().strip().splitlines()[-1]); print('[new lib, reserve $r more]', round(d['ms_per_step'],3))"
done
