mkdir -p gpurun_out/r4h
R=$(pwd)
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace -d /tmp/p_slow -o t -- python3 $R/tools/ddp_slow_trace.py > $R/gpurun_out/r4h/slow.log 2>&1
grep -v "Gloo\|socket\|amdgpu.ids\|simple_timer" $R/gpurun_out/r4h/slow.log | tail -20
python3 $R/tools/timeline.py /tmp/p_slow/t_results.db $R/gpurun_out/r4h/slow
head -60 $R/gpurun_out/r4h/slow_summary.md
echo done
