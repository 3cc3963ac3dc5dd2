# round 4, call B: dispatch-order probe; ctrl_vl-bert_base (100 regions) kernel stats; LayerNorm TA / TCP counters (item 7)
set -e
R=$(pwd)
OUT=$R/gpurun_out/r4b
mkdir -p $OUT
./tools/micro/dispatch_order > $OUT/dispatch_order.txt 2>&1 || true
cat $OUT/dispatch_order.txt
cd /tmp && export TMPDIR=/tmp
rocprofv3 -L > $OUT/counters.txt 2>&1 || true
B="$R/bench.py --no-cpu-baseline --no-kernel-timing"
rocprofv3 --kernel-trace -d /tmp/p_vl -o t -- python3 $B --config ctrl_vl-bert_base --regions 100 --steps 4 --warmup 2 --serial > $OUT/vlbert.log 2>&1
python3 $R/tools/rocprof_summary.py /tmp/p_vl/t_results.db $OUT/vlbert_r100_kernel_stats
rm -rf /tmp/p_vl
echo "vlbert trace done" >> $OUT/progress.txt
for set in "TA_BUSY_sum TA_TA_BUSY_sum" "TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES SQ_WAIT_ANY"; do
  tag=$(echo $set | tr ' ' '+')
  rocprofv3 --kernel-trace --pmc $set -d /tmp/p_ln -o t -- python3 $B --steps 1 --warmup 1 --serial > $OUT/pmc_$tag.log 2>&1 || { echo "pmc $tag failed" >> $OUT/progress.txt; tail -5 $OUT/pmc_$tag.log; continue; }
  python3 $R/tools/rocprof_summary.py /tmp/p_ln/t_results.db $OUT/pmc_$tag --pmc X || true
  rm -rf /tmp/p_ln
  echo "pmc $tag done" >> $OUT/progress.txt
done
echo done
