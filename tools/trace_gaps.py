"""Inter-kernel gaps per hardware queue in a rocprofv3 kernel trace: how much of a step is spent between kernels?
usage: trace_gaps.py results.db [n_last_dispatches_to_analyse]"""
import sqlite3, sys, collections
c = sqlite3.connect(sys.argv[1])
tabs = [r[0] for r in c.execute("select name from sqlite_master where type='table'")]
kd = [t for t in tabs if "kernel_dispatch" in t][0]; ks = [t for t in tabs if "kernel_symbol" in t][0]
rows = list(c.execute(f"select s.kernel_name, d.start, d.end, d.queue_id from {kd} d join {ks} s on d.kernel_id = s.id order by d.start"))
n = int(sys.argv[2]) if len(sys.argv) > 2 else len(rows) // 3
rows = rows[-n:]
span = (rows[-1][2] - rows[0][1]) / 1e6
byq = collections.defaultdict(list)
for r in rows:
    byq[r[3]].append(r)
print("analysed %d dispatches over %.3f ms" % (len(rows), span))
for q, rs in sorted(byq.items(), key=lambda kv: -len(kv[1])):
    busy = sum(e - s for _, s, e, _ in rs) / 1e6
    gaps = [max(0, rs[i + 1][1] - rs[i][2]) for i in range(len(rs) - 1)]
    small = [g for g in gaps if g < 20000]
    print("queue %s: %d kernels, busy %.3f ms, gaps < 20 us: %d totalling %.3f ms (median %.2f us), larger gaps totalling %.3f ms" % (
        q, len(rs), busy, len(small), sum(small) / 1e6, sorted(small)[len(small) // 2] / 1e3 if small else 0.0, (sum(gaps) - sum(small)) / 1e6))
