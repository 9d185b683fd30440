"""Per-kernel statistics of a rocprofv3 --kernel-trace database restricted to the launches AFTER the last launch of a marker kernel (tools/proof_only.py puts
k_merkle_paths between Context::generate + warm-up and the measured proofs): the launch population bench.py's HIP-event profile times.
usage: python tools/r04/stats_after_marker.py <x_results.db> <marker kernel> <out.csv> "<comment>" """
import csv, os, sqlite3, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from rocpd_summary import short
db = sqlite3.connect(sys.argv[1])
rows = db.execute("select name, start, end from kernels order by start").fetchall()
names = [short(r[0]) for r in rows]
last = max(i for i, n in enumerate(names) if n.split(":", 1)[-1].startswith(sys.argv[2]))
agg = {}
for n, r in list(zip(names, rows))[last + 1:]:
    a = agg.setdefault(n, [0, 0, 1 << 62, 0])
    d = r[2] - r[1]
    a[0] += 1; a[1] += d; a[2] = min(a[2], d); a[3] = max(a[3], d)
tot = sum(a[1] for a in agg.values())
with open(sys.argv[3], "w", newline="") as f:
    f.write(f"\"# {sys.argv[4]}\"\n")
    w = csv.writer(f)
    w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "MinNs", "MaxNs", "Percentage"])
    for n, a in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        w.writerow([n, a[0], a[1], round(a[1] / a[0], 1), a[2], a[3], round(100.0 * a[1] / tot, 3)])
for n, a in sorted(agg.items(), key=lambda kv: -kv[1][1])[:8]:
    print(f"{n:36s} calls {a[0]:5d} avg {a[1] / a[0] / 1e3:9.1f} us  {100.0 * a[1] / tot:5.1f} %")
