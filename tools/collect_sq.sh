# SQ wait / issue breakdown per kernel of one step (rocprofv3 --pmc, SQ block only; separate from the kernel-trace-only and TCC passes).
set -e
R=$(pwd)
OUT=$R/gpurun_out/prof_sq
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
B="$R/bench.py --no-cpu-baseline --no-kernel-timing --steps 1 --warmup 1 --serial"
rocprofv3 -L > $OUT/counters.txt 2>&1 || true
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY -d /tmp/p_sq1 -o t -- python3 $B > $OUT/sq1.log 2>&1
python3 $R/tools/rocprof_summary.py /tmp/p_sq1/t_results.db $OUT/pmc_sq_wait --pmc SQ
rm -rf /tmp/p_sq1
echo "pass 1 done" >> $OUT/progress.txt
rocprofv3 --kernel-trace --pmc SQ_BUSY_CYCLES SQ_WAVES SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_INSTS_LDS -d /tmp/p_sq2 -o t -- python3 $B > $OUT/sq2.log 2>&1
python3 $R/tools/rocprof_summary.py /tmp/p_sq2/t_results.db $OUT/pmc_sq_insts --pmc SQ
rm -rf /tmp/p_sq2
echo "pass 2 done" >> $OUT/progress.txt
