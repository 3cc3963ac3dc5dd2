mkdir -p gpurun_out/r4f
timeout -k 10 400 python3 tools/ddp_overhead.py > gpurun_out/r4f/ddp_overhead.txt 2>&1
cat gpurun_out/r4f/ddp_overhead.txt
R=$(pwd)
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d /tmp/p_v -o t -- python3 $R/tools/vendor_kernel_names.py > $R/gpurun_out/r4f/vendor.log 2>&1
python3 $R/tools/rocprof_summary.py /tmp/p_v/t_results.db $R/gpurun_out/r4f/vendor_kernels
cat $R/gpurun_out/r4f/vendor_kernels.md
echo done
