"""profiles/r01_pmc_hbm_traffic.md + r01_pmc_summary.json from the FETCH_SIZE / WRITE_SIZE csv summaries (tools/rocprof_summary.py --pmc).
usage: pmc_traffic.py <fetch.csv> <write.csv> <steps profiled> <out_md> <out_json>"""
import csv, json, sys

fetch, write, steps, out_md, out_json = sys.argv[1], sys.argv[2], int(sys.argv[3]), sys.argv[4], sys.argv[5]
rows = {}
for path, key in ((fetch, "f"), (write, "w")):
    for r in csv.DictReader(open(path)):
        d = rows.setdefault(r["kernel"], {"n": 0, "f": 0.0, "w": 0.0})
        d["n"] = max(d["n"], int(r["dispatches"]))
        d[key] += float(r["sum"])
# rocprofv3 reports both counters in KB; gfx950 counts a 128-byte fetch request as 64 bytes (MI355X_MICROARCH.md, HBM section): x2
hbm = lambda d: (2.0 * d["f"] + d["w"]) * 1024.0
with open(out_md, "w") as md:
    md.write("# rocprofv3 --kernel-trace --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-kernel-timing --serial\n\n")
    md.write("Counter sums are in KB as rocprofv3 reports them; FETCH_SIZE is doubled in the corrected column (gfx950 counts a 128-byte request as 64 bytes, MI355X_MICROARCH.md HBM section).  %d steps profiled.\n\n" % steps)
    md.write("| kernel | launches (%d steps) | FETCH_SIZE sum KB (raw) | WRITE_SIZE sum KB | corrected HBM MB / launch |\n|---|---|---|---|---|\n" % steps)
    for k, d in sorted(rows.items(), key=lambda kv: -hbm(kv[1])):
        if hbm(d) / 1e6 < 20:
            continue
        md.write("| %s | %d | %.0f | %.0f | %.1f |\n" % (k[:84], d["n"], d["f"], d["w"], hbm(d) / d["n"] / 1e6))
g = [d for k, d in rows.items() if "gemm" in k]
n = sum(d["n"] for d in g) / steps
fb, wb = sum(2.0 * d["f"] * 1024 for d in g) / steps, sum(d["w"] * 1024 for d in g) / steps
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
json.dump({"source_fingerprint": bench.source_fingerprint(), "kernel": "vk::gemm256k_kernel / gemm256p_kernel / gemm_kernel family", "launches_per_step": n, "fetch_bytes_per_step": fb, "write_bytes_per_step": wb,
           "hbm_bytes_per_launch": (fb + wb) / n,
           "note": "rocprofv3 --kernel-trace --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes over `bench.py --steps 1 --warmup 1 --serial`; KB units x1024; FETCH_SIZE doubled (gfx950 counts 128-B requests as 64 B)"},
          open(out_json, "w"), indent=1)
print("gemm launches/step", n, "HBM MB per launch", (fb + wb) / n / 1e6)
