#!/usr/bin/env python
"""Is the chip out of work, or out of issue slots? A rocprofv3 --kernel-trace of proofs in flight, read as a TIMELINE of what all cohort queues run at the same instant
(trace_analyze.py looks at one queue at a time). usage:
  python tools/timeline_occupancy.py <x_results.db> [bin_ms]
Classes of a cohort launch (kc:*), by what it can keep busy:
  H  wide hash layers: k_merkle_layer / k_merkle_leaves* with >= 128 workgroups over all members (VALU bound: one compress per lane; a merged launch is capped at 256)
  h  narrower hash launches (k_merkle_layer below that, k_merkle_layer_lp, k_merkle_tail)
  W  other wide launches (>= 1024 workgroups): the streaming kernels of the batch opening and the commits
  w  other launches of 2 .. 1023 workgroups per member set
  T  one-workgroup-per-member protocol tails (k_*_tail, k_sc_*), latency bound: a sponge wave each
  .  nothing of the queue on the device (host phase, or the gap between two launches)
Printed: (1) the share of the measured batch's wall time by the number of queues in class H at the instant, and what the OTHER queues run when no H is active;
(2) a character timeline, one row per queue, one column per bin (the class that holds most of the bin), over the middle of the measured batch: cohorts in phase
show as vertical stripes."""
import collections
import sqlite3
import sys

sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.abspath(__file__)))
from rocpd_summary import short


def klass(name, wgs):
    k = short(name)[3:]
    if "merkle_layer_lp" in k or "merkle_tail" in k:
        return "h"
    if "merkle_layer" in k or "merkle_leaves" in k:
        return "H" if wgs >= 128 else "h"  # (round 6 caps every merged launch at 256 workgroups: a hash layer that takes its whole cap is a wide one)
    if any(t in k for t in ("_tail", "sc_persist", "sc_small")):
        return "T"
    return "W" if wgs >= 1024 else "w"


def main():
    db = sqlite3.connect(sys.argv[1])
    bin_ms = float(sys.argv[2]) if len(sys.argv) > 2 else 4.0
    rows = db.execute("select name, start, end, grid_x, grid_y, grid_z, workgroup_x, queue_id from kernels order by start").fetchall()
    kc = [r for r in rows if short(r[0]).startswith("kc:")]
    if not kc:
        print("no cohort launches (kc:*) in this trace")
        return
    queues = sorted(set(r[7] for r in kc))
    # the measured batch = the second half of every queue's launches
    half = []
    for q in queues:
        rq = [r for r in kc if r[7] == q]
        half += rq[len(rq) // 2:]
    t0, t1 = min(r[1] for r in half), max(r[2] for r in half)
    wg = lambda r: max(1, r[3] // max(1, r[6])) * max(1, r[4]) * max(1, r[5])  # noqa: E731
    # (1) sweep over the launch boundaries
    ev = []
    for r in half:
        c = klass(r[0], wg(r))
        ev.append((r[1], 1, c)); ev.append((r[2], -1, c))
    ev.sort()
    act = collections.Counter()
    last = t0
    by_h = collections.Counter()          # time by number of H launches active
    rest_when_no_h = collections.Counter()  # time-weighted count of the other classes while no H is active
    any_wide = 0
    nothing_wide_or_hash = 0
    for t, d, c in ev:
        dt = t - last
        if dt > 0:
            by_h[min(act["H"], 6)] += dt
            if act["H"] == 0:
                for k in "hWwT":
                    rest_when_no_h[k] += dt * act[k]
                if act["W"] == 0:
                    nothing_wide_or_hash += dt
            if act["H"] or act["W"]:
                any_wide += dt
        act[c] += d
        last = t
    span = t1 - t0
    print(f"measured batch: {len(half)} cohort launches on {len(queues)} queues over {span / 1e6:.1f} ms")
    print("share of the wall time by the number of chip-filling hash launches (H) active at the instant:")
    for k in sorted(by_h):
        print(f"   {k if k < 6 else '6+'} active: {100.0 * by_h[k] / span:5.1f} %")
    nh = by_h[0] or 1
    print(f"while no H is active ({100.0 * by_h[0] / span:.1f} % of the time) the queues run on average: " + ", ".join(f"{rest_when_no_h[k] / nh:.1f} {k}" for k in "hWwT") +
          f"; of that time {100.0 * nothing_wide_or_hash / nh:.1f} % has no wide launch (W) either")
    print(f"some chip-filling launch (H or W) is active {100.0 * any_wide / span:.1f} % of the wall time")
    # (2) timeline
    nb = int(span / (bin_ms * 1e6)) + 1
    print(f"timeline, {bin_ms:g} ms per column, one row per queue (H/h hash wide/narrow, W/w other wide/narrow, T tails, . idle):")
    for q in queues:
        occ = [collections.Counter() for _ in range(nb)]
        for r in half:
            if r[7] != q:
                continue
            c = klass(r[0], wg(r))
            a, b = r[1] - t0, r[2] - t0
            i = int(a / (bin_ms * 1e6))
            while i < nb and i * bin_ms * 1e6 < b:
                lo, hi = max(a, i * bin_ms * 1e6), min(b, (i + 1) * bin_ms * 1e6)
                occ[i][c] += hi - lo
                i += 1
        line = ""
        for i in range(nb):
            tot = sum(occ[i].values())
            if tot < 0.5 * bin_ms * 1e6:
                line += "."
            else:
                line += occ[i].most_common(1)[0][0]
        print("   " + line[:420])


if __name__ == "__main__":
    main()
