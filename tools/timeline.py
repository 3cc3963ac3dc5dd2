"""One step's kernel timeline out of a rocprofv3 kernel trace (results.db): per hardware queue the launches in start order with the gap
in front of each, the union of busy time over all queues and the time in which only one / two queues are busy.
usage: timeline.py results.db out_prefix [step_marker_kernel]
The step is cut at consecutive launches of `step_marker_kernel` (default: the first embedding kernel of the forward list)."""
import collections
import sqlite3
import sys

db, out = sys.argv[1], sys.argv[2]
marker = sys.argv[3] if len(sys.argv) > 3 else "mask_prep"
c = sqlite3.connect(db)
tabs = [r[0] for r in c.execute("select name from sqlite_master where type='table'")]
kd = [t for t in tabs if "kernel_dispatch" in t][0]
ks = [t for t in tabs if "kernel_symbol" in t][0]
rows = list(c.execute(f"select s.kernel_name, d.start, d.end, d.queue_id, d.grid_size_x, d.workgroup_size_x from {kd} d join {ks} s on d.kernel_id = s.id order by d.start"))
marks = [i for i, r in enumerate(rows) if marker in r[0]]
# mask_prep is launched twice per forward (text, vision): a step begins at every second one
starts = marks[::2] if len(marks) >= 4 else marks
if len(starts) < 3:
    raise SystemExit("fewer than three steps in the trace")
i0, i1 = starts[-2], starts[-1]
step = rows[i0:i1]
t0 = step[0][1]
span = (rows[i1][1] - t0) / 1e3


def short(n):
    n = n.replace("vk::", "")
    return n[:n.index("(")] if "(" in n else n[:70]


byq = collections.defaultdict(list)
for r in step:
    byq[r[3]].append(r)
with open(out + "_timeline.csv", "w") as fh:
    fh.write("queue,start_us,dur_us,gap_before_us,wgs,kernel\n")
    for q, rs in byq.items():
        prev = None
        for n, s, e, _, gx, wx in rs:
            fh.write("%s,%.2f,%.2f,%.2f,%d,%s\n" % (q, (s - t0) / 1e3, (e - s) / 1e3, (s - prev) / 1e3 if prev else 0.0, gx // max(wx, 1), short(n)))
            prev = e
# union / overlap
ev = []
for n, s, e, q, _, _ in step:
    ev.append((s, 1))
    ev.append((e, -1))
ev.sort()
depth, last, busy = 0, t0, collections.Counter()
for t, d in ev:
    busy[depth] += t - last
    last = t
    depth += d
busy[0] += rows[i1][1] - last
with open(out + "_summary.md", "w") as fh:
    fh.write("step of %.1f us, %d launches on %d queues\n\n" % (span, len(step), len(byq)))
    fh.write("| kernels running at once | us | share |\n|---|---|---|\n")
    for k in sorted(busy):
        fh.write("| %d | %.1f | %.1f %% |\n" % (k, busy[k] / 1e3, 100.0 * busy[k] / 1e3 / span))
    fh.write("\n| queue | launches | busy us | gaps < 20 us (count, total us, median) | larger gaps us |\n|---|---|---|---|---|\n")
    for q, rs in sorted(byq.items(), key=lambda kv: -len(kv[1])):
        b = sum(e - s for _, s, e, _, _, _ in rs) / 1e3
        gaps = [max(0, rs[i + 1][1] - rs[i][2]) / 1e3 for i in range(len(rs) - 1)]
        small = sorted(g for g in gaps if g < 20)
        fh.write("| %s | %d | %.1f | %d, %.1f, %.2f | %.1f |\n" % (q, len(rs), b, len(small), sum(small), small[len(small) // 2] if small else 0, sum(gaps) - sum(small)))
    agg = collections.defaultdict(lambda: [0, 0.0])
    for n, s, e, q, _, _ in step:
        a = agg[(q, short(n))]
        a[0] += 1
        a[1] += (e - s) / 1e3
    fh.write("\n| queue | kernel | launches | total us |\n|---|---|---|---|\n")
    for (q, n), (cnt, tot) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:40]:
        fh.write("| %s | %s | %d | %.1f |\n" % (q, n, cnt, tot))
print(open(out + "_summary.md").read())
