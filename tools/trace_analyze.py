#!/usr/bin/env python
"""What a rocprofv3 --kernel-trace of proofs in flight says about the cohort regime. usage:
  python tools/trace_analyze.py <x_results.db> [--sequence]
* per-queue busy fraction: sum of (end - start) of a queue's kernels / the queue's span — in these traces the next kernel
  of a queue starts the instant its predecessor ends, so (end - start) includes the wait for execution resources;
* duration distribution of the one-wave k_publish (3 us alone): the price of a dispatch under load;
* CUs held exclusively: workgroups of kernels launched with >= 80 KB of dynamic LDS (the CU reservation of the
  one-workgroup kernels), resident by the trace's timestamps — time average, quantiles, CU-seconds per kernel;
* --sequence: the run-length compressed launch sequence of the last proof step of one cohort queue with durations (us)."""
import sqlite3
import sys

sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.abspath(__file__)))
from rocpd_summary import short


def quantiles(vals, qs):
    v = sorted(vals)
    return [v[min(len(v) - 1, int(q * len(v)))] for q in qs] if v else []


def main():
    db = sqlite3.connect(sys.argv[1])
    rows = db.execute("select name, start, end, grid_x, grid_y, grid_z, workgroup_x, queue_id, lds_size from kernels order by start").fetchall()
    kc = [r for r in rows if short(r[0]).startswith("kc:")]
    if not kc:
        print("no cohort launches (kc:*) in this trace")
        return
    t0, t1 = kc[0][1], max(r[2] for r in kc)
    span = (t1 - t0) / 1e9
    print(f"{len(kc)} cohort launches on {len(set(r[7] for r in kc))} queues over {span:.3f} s")
    busy = []
    for q in sorted(set(r[7] for r in kc)):
        rq = [r for r in kc if r[7] == q]
        busy.append(sum(r[2] - r[1] for r in rq) / max(1, rq[-1][2] - rq[0][1]))
    print("queue busy fraction (kernel start..end / queue span): min %.2f  mean %.2f  max %.2f" % (min(busy), sum(busy) / len(busy), max(busy)))
    pub = [(r[2] - r[1]) / 1e3 for r in kc if short(r[0]) == "kc:k_publish"]
    if pub:
        print("k_publish start..end (us): p5 %.1f  p50 %.1f  p75 %.1f  p95 %.1f  max %.1f" % tuple(quantiles(pub, (0.05, 0.5, 0.75, 0.95, 0.999999))))
    excl = [r for r in kc if r[8] >= 80000]
    wg = lambda r: (r[3] // r[6]) * r[4] * r[5]  # noqa: E731
    cu_s = sum((r[2] - r[1]) / 1e9 * wg(r) for r in excl)
    print(f"exclusive workgroups: {cu_s:.1f} CU-seconds -> {cu_s / span:.1f} CUs held on average")
    by = {}
    for r in excl:
        by[short(r[0])] = by.get(short(r[0]), 0.0) + (r[2] - r[1]) / 1e9 * wg(r)
    for k, v in sorted(by.items(), key=lambda kv: -kv[1]):
        print(f"   {k:40s} {v:8.2f} CU-s")
    ev = sorted([(r[1], wg(r)) for r in excl] + [(r[2], -wg(r)) for r in excl])
    hist, cur, last = {}, 0, ev[0][0] if ev else 0
    for t, d in ev:
        hist[cur] = hist.get(cur, 0) + (t - last)
        last, cur = t, cur + d
    tot, acc, qs = sum(hist.values()) or 1, 0, {}
    for k in sorted(hist):
        acc += hist[k]
        for q in (0.5, 0.9, 0.99):
            if q not in qs and acc >= q * tot:
                qs[q] = k
    print("exclusive workgroups resident (trace timestamps): p50 %s  p90 %s  p99 %s  max %s" % (qs.get(0.5), qs.get(0.9), qs.get(0.99), max(hist) if hist else 0))
    # --- round 2: where does a cohort's chain spend its time? (second half of every queue = the measured batch)
    import collections, statistics
    import numpy as np
    half = []
    for q in sorted(set(r[7] for r in kc)):
        rq = [r for r in kc if r[7] == q]
        half += rq[len(rq) // 2:]
    nq = len(set(r[7] for r in half))
    gap = 0
    for q in set(r[7] for r in half):
        rq = [r for r in half if r[7] == q]
        gap += sum(max(0, b[1] - a[2]) for a, b in zip(rq, rq[1:]))
    span_q = sum(max(r[2] for r in half if r[7] == q) - min(r[1] for r in half if r[7] == q) for q in set(r[7] for r in half)) / nq
    print(f"measured batch: {len(half) // nq} launches per queue, queue span {span_q / 1e6:.1f} ms, host gaps {gap / nq / 1e6:.1f} ms per queue")
    cls = collections.defaultdict(lambda: [0, 0.0])
    for r in half:
        k = short(r[0])[3:]
        c = "one-workgroup tails" if any(t in k for t in ("_tail", "sc_persist", "sc_small")) else "merkle layers" if "merkle_layer" in k else "other"
        cls[c][0] += 1; cls[c][1] += (r[2] - r[1]) / 1e6
    for c, (cnt, ms) in sorted(cls.items(), key=lambda kv: -kv[1][1]):
        print(f"   {c:22s} {cnt / nq:6.1f} launches  {ms / nq:8.1f} ms per queue")
    starts = np.array(sorted(r[1] for r in kc)); ends = np.array(sorted(r[2] for r in kc))
    small = [r for r in half if short(r[0]) in ("kc:k_publish", "kc:k_copy_words", "kc:k_fold", "kc:k_reduce_publish", "kc:k_bf_msg", "kc:k_fri_fold")]
    by = collections.defaultdict(list)
    for r in small:
        act = int(np.searchsorted(starts, r[1], "right") - np.searchsorted(ends, r[1], "right"))
        by[min(28, act) // 4 * 4].append((r[2] - r[1]) / 1e3)
    print("duration of the SMALL kernels (publish, copy_words, fold, ...) by the number of kernels active on all queues when they start:")
    for k in sorted(by):
        print(f"   ~{k:2d} active: n = {len(by[k]):5d}  median {statistics.median(by[k]):8.1f} us")
    # --- what slows the one-workgroup protocol kernels down under load? Their duration against the number of WIDE kernels
    # (>= 512 workgroups) running meanwhile (16 samples inside every instance), with the solo duration (kg:*, the single proof
    # of the bench's latency measurement) for reference
    wg_total = lambda r: max(1, r[3] // max(1, r[6])) * r[4] * r[5]  # noqa: E731
    wide = [r for r in rows if wg_total(r) >= 512]
    ws = np.array(sorted(r[1] for r in wide)); we = np.array(sorted(r[2] for r in wide))
    tails = [r for r in half if any(t in short(r[0]) for t in ("_tail", "sc_persist"))]
    solo = collections.defaultdict(list)
    for r in rows:
        if short(r[0]).startswith("kg:") and any(t in short(r[0]) for t in ("_tail", "sc_persist")):
            solo[short(r[0])[3:]].append((r[2] - r[1]) / 1e3)
    tab = collections.defaultdict(lambda: collections.defaultdict(list))
    edges = (0.05, 0.5, 1.5, 3.0)
    for r in tails:
        ts = np.linspace(r[1], r[2], 18)[1:-1]
        act = float(np.mean(np.searchsorted(ws, ts, "right") - np.searchsorted(we, ts, "right")))
        b = sum(act > e for e in edges)
        tab[short(r[0])[3:]][b].append((r[2] - r[1]) / 1e3)
    print("one-workgroup kernels: median duration (us) by the average number of wide kernels (>= 512 workgroups) running meanwhile")
    print("   %-28s %9s | %14s %14s %14s %14s %14s" % ("kernel", "solo", "none", "<0.5", "0.5-1.5", "1.5-3", ">3"))
    for k in sorted(tab, key=lambda k: -sum(sum(v) for v in tab[k].values())):
        cells = []
        for b in range(5):
            v = tab[k].get(b, [])
            cells.append("%8.0f (%4d)" % (statistics.median(v), len(v)) if v else "       - (   0)")
        sv = solo.get(k, [])
        print("   %-28s %9s | %s" % (k[:28], "%.0f" % statistics.median(sv) if sv else "-", " ".join(cells)))
    if "--sequence" in sys.argv:
        q = kc[-1][7]
        seq = [r for r in kc if r[7] == q]
        seq = seq[len(seq) // 2:]
        out = []
        for r in seq:
            k, d = short(r[0])[3:].replace("k_", ""), (r[2] - r[1]) / 1e3
            if out and out[-1][0] == k:
                out[-1][1] += 1; out[-1][2] += d
            else:
                out.append([k, 1, d])
        line = ""
        for k, c, d in out:
            item = f"{k}{'x' + str(c) if c > 1 else ''}({d:.0f})"
            if len(line) + len(item) > 150:
                print(line); line = ""
            line += item + " "
        print(line)
        print("this half of the queue: wall %.1f ms, kernel start..end %.1f ms" % ((seq[-1][2] - seq[0][1]) / 1e6, sum(r[2] - r[1] for r in seq) / 1e6))


if __name__ == "__main__":
    main()
