"""Turn a rocprofv3 results .db (kernel trace and/or PMC) into the markdown / csv summaries kept under profiles/.
Usage: rocprof_summary.py <results.db> <out_prefix> [--pmc COUNTER]"""
import sqlite3, sys, collections, json, re


def short(name):
    n = re.sub(r"^_ZN2vk\d+", "vk::", name.split("(")[0])
    n = n.replace(".kd", "")
    return n[:96]


def main():
    db, prefix = sys.argv[1], sys.argv[2]
    c = sqlite3.connect(db)
    tabs = [r[0] for r in c.execute("select name from sqlite_master where type='table'")]
    kd = [t for t in tabs if "kernel_dispatch" in t][0]
    ks = [t for t in tabs if "kernel_symbol" in t][0]
    if "--pmc" in sys.argv:
        pe = [t for t in tabs if t.startswith("rocpd_pmc_event")][0]
        pi = [t for t in tabs if t.startswith("rocpd_info_pmc")][0]
        q = (f"select s.kernel_name, i.name, count(distinct d.id), sum(e.value) from {pe} e join {pi} i on e.pmc_id = i.id "
             f"join {kd} d on e.event_id = d.event_id join {ks} s on d.kernel_id = s.id group by s.kernel_name, i.name")
        agg = collections.defaultdict(lambda: [0, 0.0])
        for name, ctr, n, v in c.execute(q):
            k = (short(name), ctr)
            agg[k][0] += n
            agg[k][1] += v
        with open(prefix + ".csv", "w") as fh:
            fh.write("kernel,counter,dispatches,sum\n")
            for (k, ctr), (n, v) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
                fh.write('"%s",%s,%d,%.0f\n' % (k, ctr, n, v))
        print("wrote", prefix + ".csv", len(agg), "rows")
        return
    # (the one-wave hold kernels are the start-up stream probe of volta_amd/streams.py -- thousands of 1 us launches before the first step --
    # not part of a step: left out of the table)
    rows = [r for r in c.execute(f"select s.kernel_name, d.start, d.end from {kd} d join {ks} s on d.kernel_id = s.id") if "hold_cus_kernel" not in r[0]]
    agg = collections.defaultdict(list)
    for n, s, e in rows:
        agg[short(n)].append(e - s)
    tot = sum(sum(v) for v in agg.values())
    with open(prefix + ".csv", "w") as fh, open(prefix + ".md", "w") as md:
        fh.write("kernel,calls,total_ns,avg_ns,min_ns,max_ns,percent\n")
        md.write("| kernel | calls | total ms | avg us | min us | max us | % |\n|---|---|---|---|---|---|---|\n")
        for k, v in sorted(agg.items(), key=lambda kv: -sum(kv[1])):
            fh.write('"%s",%d,%d,%.0f,%d,%d,%.2f\n' % (k, len(v), sum(v), sum(v) / len(v), min(v), max(v), 100.0 * sum(v) / tot))
            if sum(v) / tot > 0.002:
                md.write("| %s | %d | %.3f | %.1f | %.1f | %.1f | %.2f |\n" % (k, len(v), sum(v) / 1e6, sum(v) / len(v) / 1e3, min(v) / 1e3, max(v) / 1e3, 100.0 * sum(v) / tot))
    print("wrote", prefix + ".md/.csv", len(agg), "kernels, total %.3f ms" % (tot / 1e6))


if __name__ == "__main__":
    main()
