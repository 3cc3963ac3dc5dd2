# Step time of every BASELINE configuration on one GPU (writes gpurun_out/cfg_*.json, prints a summary line each).
set -e
mkdir -p gpurun_out
run() { tag=$1; shift; timeout -k 10 250 python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-kernel-timing "$@" > gpurun_out/cfg_$tag.json 2> gpurun_out/cfg_$tag.err; python3 -c "
import json,sys; d=json.loads(open('gpurun_out/cfg_$tag.json').read().strip().splitlines()[-1]); print('$tag', round(d['ms_per_step'],2), 'ms', round(d['value']), 'pairs/s', 'frac', round(d.get('roofline',{}).get('frac',0),4))"; }
run vilbert_b256
run vilbert_b512 --batch 512
run vilbert_t38 --seq-len 38
run lxmert_b256 --config ctrl_lxmert
run uniter_b512 --config ctrl_uniter_base --batch 512
run uniter_b256 --config ctrl_uniter_base
run visualbert_b256 --config ctrl_visualbert_base
run vlbert_b256 --config ctrl_vl-bert_base
run vlbert_r100_b256 --config ctrl_vl-bert_base --regions 100
run vlbert_r100_b256_fp8 --config ctrl_vl-bert_base --regions 100 --dtype fp8
run uniter_b512_fp8 --config ctrl_uniter_base --batch 512 --dtype fp8
# the reference's non-ctrl configurations: original-paper geometries and targets
run vilbert_base_b256 --config vilbert_base
run lxmert_orig_b256 --config lxmert
run vlbert_orig_b256 --config vl-bert_base
