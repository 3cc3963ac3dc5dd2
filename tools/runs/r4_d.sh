mkdir -p gpurun_out/r4d
timeout -k 10 300 python3 tools/bench_soft_pair.py > gpurun_out/r4d/soft_pair.txt 2>&1
cat gpurun_out/r4d/soft_pair.txt
cd /tmp && export TMPDIR=/tmp
R=/root/repo
VK_SOFT=1 rocprofv3 --kernel-trace -d /tmp/p_soft -o t -- python3 $R/bench.py --no-cpu-baseline --no-kernel-timing --steps 4 --warmup 2 > $R/gpurun_out/r4d/trace_soft.log 2>&1
python3 - <<'PY'
import sqlite3
c = sqlite3.connect("/tmp/p_soft/t_results.db")
tabs = [r[0] for r in c.execute("select name from sqlite_master where type='table'")]
kd = [t for t in tabs if "kernel_dispatch" in t][0]
ks = [t for t in tabs if "kernel_symbol" in t][0]
rows = list(c.execute(f"select s.kernel_name, d.start, d.end, d.queue_id from {kd} d join {ks} s on d.kernel_id = s.id order by d.start"))
t0 = rows[0][1]
# last step only: print the kernels of a window in the dual-stream forward
n = len(rows)
out = open("/root/repo/gpurun_out/r4d/trace_soft_window.txt", "w")
for name, s, e, q in rows[int(n * 0.80):int(n * 0.80) + 400]:
    out.write("%10.1f %10.1f %7.1f q%d %s\n" % ((s - t0) / 1e3, (e - t0) / 1e3, (e - s) / 1e3, q, name[:70]))
PY
echo done
