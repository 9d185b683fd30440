#!/usr/bin/env python
"""How busy is the GPU during a traced run? From a rocprofv3 rocpd database (--kernel-trace): wall span of the kernels,
union of their intervals (busy time), per-kernel totals, mean number of kernels resident, and the same restricted to
wide launches (>= 256 workgroups). usage: python tools/trace_busy.py x_results.db [out.json]"""
import json, sqlite3, sys
db = sqlite3.connect(sys.argv[1])
cols = [r[1] for r in db.execute("pragma table_info(kernels)")]
def col(*names):
    for n in names:
        if n in cols: return n
    return None
gx, wx = col("grid_size_x", "grid_x", "grid_size"), col("workgroup_size_x", "workgroup_x", "workgroup_size")
gy, gz = col("grid_size_y", "grid_y"), col("grid_size_z", "grid_z")
sel = "name, start, end" + (f", {gx}, {wx}" if gx and wx else ", 0, 1") + (f", {gy}, {gz}" if gy and gz else ", 1, 1")
rows = db.execute(f"select {sel} from kernels order by start").fetchall()
def union(iv):
    tot, cur_s, cur_e = 0, None, None
    for s, e in iv:
        if cur_e is None or s > cur_e:
            if cur_e is not None: tot += cur_e - cur_s
            cur_s, cur_e = s, e
        elif e > cur_e: cur_e = e
    if cur_e is not None: tot += cur_e - cur_s
    return tot
span = max(r[2] for r in rows) - rows[0][1]
busy = union([(r[1], r[2]) for r in rows])
def wgs(r): return (r[3] // max(r[4], 1)) * max(r[5], 1) * max(r[6], 1)
wide = [(r[1], r[2]) for r in rows if wgs(r) >= 256]
agg = {}
for r in rows:
    a = agg.setdefault(r[0], [0, 0, 0]); a[0] += 1; a[1] += r[2] - r[1]; a[2] += wgs(r)
out = {"columns": cols, "kernels": len(rows), "span_ms": span / 1e6, "busy_ms": busy / 1e6, "busy_frac": busy / span,
       "mean_resident_kernels": sum(r[2] - r[1] for r in rows) / span, "wide_busy_ms": union(wide) / 1e6, "wide_busy_frac": union(wide) / span,
       "top": [{"kernel": k[:70], "calls": v[0], "total_ms": v[1] / 1e6, "avg_us": v[1] / v[0] / 1e3, "avg_wgs": v[2] / v[0]} for k, v in sorted(agg.items(), key=lambda kv: -kv[1][1])[:25]]}
print(json.dumps({k: v for k, v in out.items() if k not in ("top", "columns")}))
for t in out["top"]: print(f"{t['kernel']:70s} calls={t['calls']:6d} total_ms={t['total_ms']:9.2f} avg_us={t['avg_us']:9.1f} avg_wgs={t['avg_wgs']:8.1f}")
if len(sys.argv) > 2: json.dump(out, open(sys.argv[2], "w"), indent=1)

# ---- per-queue view and what a tiny kernel's duration depends on
qc = col("queue_id", "queue", "stream_id")
if qc:
    qrows = db.execute(f"select {qc}, start, end, name from kernels order by start").fetchall()
    per = {}
    for q, s, e, n in qrows: per.setdefault(q, []).append((s, e, n))
    print("per queue: kernels, busy fraction of the span, median gap between consecutive kernels (us)")
    for q, ks in sorted(per.items(), key=lambda kv: -len(kv[1]))[:10]:
        gaps = sorted(ks[i + 1][0] - ks[i][1] for i in range(len(ks) - 1))
        print(f"  queue {q}: {len(ks):6d} kernels, busy {sum(e - s for s, e, _ in ks) / span:.3f}, gap median {gaps[len(gaps) // 2] / 1e3 if gaps else 0:8.1f} p90 {gaps[int(len(gaps) * .9)] / 1e3 if gaps else 0:8.1f}")
import bisect
wide_iv = sorted(wide)
wstarts = [w[0] for w in wide_iv]
def overlaps_wide(s, e):
    i = bisect.bisect_right(wstarts, e)
    return any(w[1] > s for w in wide_iv[max(0, i - 64):i])
for pat in ("k_publish", "k_reduce_publish", "k_finish_publish", "k_merkle_layer_lp", "k_sc_small"):
    d_w, d_n = [], []
    for r in rows:
        if pat in r[0] and (pat != "k_publish" or "reduce" not in r[0] and "finish" not in r[0]):
            (d_w if overlaps_wide(r[1], r[2]) else d_n).append((r[2] - r[1]) / 1e3)
    for tag, d in (("overlapping a wide launch", d_w), ("no wide launch resident", d_n)):
        if d:
            d.sort(); print(f"  {pat:20s} {tag:26s}: n={len(d):5d} median {d[len(d) // 2]:8.1f} us  p10 {d[len(d) // 10]:8.1f}  p90 {d[int(len(d) * .9)]:8.1f}")
