"""Which kernels surround the runtime's copyBuffer / fillBuffer dispatches in a rocprofv3 kernel trace?  usage: trace_neighbors.py results.db"""
import sqlite3, sys, collections
c = sqlite3.connect(sys.argv[1])
tabs = [r[0] for r in c.execute("select name from sqlite_master where type='table'")]
kd = [t for t in tabs if "kernel_dispatch" in t][0]; ks = [t for t in tabs if "kernel_symbol" in t][0]
rows = list(c.execute(f"select s.kernel_name, d.start, d.end, d.queue_id, d.grid_size_x from {kd} d join {ks} s on d.kernel_id = s.id order by d.start"))
short = lambda n: n.replace("_ZN2vk", "").split("EvNS")[0].split("(")[0][:46]
cnt = collections.Counter()
for i, r in enumerate(rows):
    if "rocclr" in r[0]:
        prev = short(rows[i - 1][0]) if i else "-"
        nxt = short(rows[i + 1][0]) if i + 1 < len(rows) else "-"
        cnt[(short(r[0]), prev, nxt, r[3], r[4])] += 1
for k, v in cnt.most_common(25):
    print(v, k)
