#!/usr/bin/env python
"""Summarise a rocprofv3 rocpd database (--kernel-trace --stats) into a per-kernel CSV: calls, total/avg/min/max ns, %.
usage: python tools/rocpd_summary.py gpurun_out/prof/x_results.db profiles/out.csv
Kernels of this library are instantiations of two wrappers (csrc/hip_dev.hip): kg<Body> = one proof on its own stream,
kc<Body> = one launch for a whole cohort of proofs (gridDim.z = members). The Itanium names of `auto` template parameters
are beyond the image's c++filt, so the body name and its bool/int template arguments are decoded here:
"kc:k_sc_persist_lds<false>"."""
import csv
import re
import sqlite3
import sys


def short(name):
    m = re.match(r"_ZN2dp2k([cg])ITnDaXadL_ZNS_(\d+)", name)
    if not m:
        return re.sub(r"^void ", "", name).split("(")[0].replace("dp::", "")
    n = int(m.group(2))
    rest = name[m.end():]
    body, rest = rest[:n], rest[n:]
    args = []
    if rest.startswith("I"):
        rest = rest[1:]
        while True:
            a = re.match(r"L([bi])(n?\d+)E", rest)
            if not a:
                break
            args.append(("true" if a.group(2) != "0" else "false") if a.group(1) == "b" else a.group(2).replace("n", "-"))
            rest = rest[a.end():]
    return f"k{m.group(1)}:{body}" + (f"<{', '.join(args)}>" if args else "")


def main():
    db = sqlite3.connect(sys.argv[1])
    raw = db.execute("select name, count(*), sum(end-start), avg(end-start), min(end-start), max(end-start) from kernels group by name").fetchall()
    agg = {}
    for name, c, tot, avg, mn, mx in raw:
        k = short(name)
        a = agg.setdefault(k, [0, 0, 1 << 62, 0])
        a[0] += c; a[1] += tot; a[2] = min(a[2], mn); a[3] = max(a[3], mx)
    rows = sorted(((k, a[0], a[1], a[1] / a[0], a[2], a[3]) for k, a in agg.items()), key=lambda r: -r[2])
    tot = sum(r[2] for r in rows) or 1
    with open(sys.argv[2], "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "MinNs", "MaxNs", "Percentage"])
        for r in rows:
            w.writerow([r[0], r[1], int(r[2]), round(r[3], 1), int(r[4]), int(r[5]), round(100.0 * r[2] / tot, 3)])
    for r in rows[:16]:
        print(f"{r[0][:60]:60s} calls={r[1]:6d} total_ms={r[2]/1e6:9.3f} avg_us={r[3]/1e3:9.2f} {100.0*r[2]/tot:5.1f}%")
    span = db.execute("select min(start), max(end) from kernels").fetchone()
    print(f"{sum(r[1] for r in rows)} launches, kernel time {tot/1e6:.1f} ms over a span of {(span[1]-span[0])/1e6:.1f} ms")


if __name__ == "__main__":
    main()
