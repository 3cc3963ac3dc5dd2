mkdir -p gpurun_out/r4z
for i in 1 2; do
for c in "0 9" "16 7" "16 6" "16 5" "8 7"; do
set -- $c
VK_OPT_CUS=$1 VK_OPT_NARROW_FROM=$2 timeout -k 10 300 python3 bench.py --no-cpu-baseline --no-kernel-timing > gpurun_out/r4z/b.json 2> gpurun_out/r4z/b.err || exit 1
python3 -c "
import json; d=json.loads(open('gpurun_out/r4z/b.json').read().strip().splitlines()[-1]); print('[VK_OPT_CUS=$1 from range $2]', round(d['ms_per_step'],3))"
done
done
