#!/usr/bin/env python
"""Summarise two rocprofv3 PMC passes (separate FETCH_SIZE and WRITE_SIZE runs of the same command, rocpd databases)
into the JSON bench.py reads for `roofline.traffic`.
usage: python tools/pmc_summary.py [--after-marker KERNEL] [--population NAME] [--units N] <fetch.db> <write.db> <out.json> "<command>" [kernel-prefix ...]
--after-marker: only launches AFTER the last launch of KERNEL count (tools/proof_only.py puts k_merkle_paths between setup and the
measured proofs); --population: a name for the set of launches summarised — bench.py only quotes a traffic figure whose population
matches the launches it timed.
Per kernel: the mean over all its launches (`hbm_bytes_per_launch`, what bench.py's per-launch `achieved` is compared with)
and the launch with the largest traffic (a sumcheck halves its tables every round, so the first launch is the big one).
Correction (MI355X_MICROARCH.md, HBM section): on gfx950 FETCH_SIZE under-reports 16 B/lane coalesced streaming reads by
2x -> doubled; WRITE_SIZE as reported; both are in KB."""
import json
import re
import sqlite3
import sys


sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.abspath(__file__)))
from rocpd_summary import short as _short  # decodes the kg<Body> / kc<Body> wrapper names
from srchash import source_sha16


def short(name):
    s = _short(name)
    return s.split(":", 1)[1] if s[:3] in ("kg:", "kc:") else s


def per_kernel(path, counter, marker=None):
    db = sqlite3.connect(path)
    rows = [(short(name), float(value), float(dur)) for name, value, dur in
            db.execute("select kernel_name, value, duration from counters_collection where counter_name = ? order by id", (counter,))]
    if marker:
        last = max((i for i, r in enumerate(rows) if r[0].split("<")[0] == marker), default=None)
        if last is None:
            raise SystemExit(f"pmc_summary: marker kernel {marker} not found in {path}")
        rows = rows[last + 1:]
    out = {}
    for name, value, dur in rows:
        out.setdefault(name, []).append((value, dur))
    return out


def main():
    argv = sys.argv[1:]
    marker = population = None
    units = None
    while argv and argv[0].startswith("--"):
        if argv[0] == "--after-marker":
            marker = argv[1]
        elif argv[0] == "--population":
            population = argv[1]
        elif argv[0] == "--units":  # proofs / sumchecks the summarised launches belong to (bench.py compares launches per unit)
            units = int(argv[1])
        else:
            raise SystemExit(f"pmc_summary: unknown option {argv[0]}")
        argv = argv[2:]
    fetch_db, write_db, out_path, command = argv[:4]
    prefixes = argv[4:]
    f, w = per_kernel(fetch_db, "FETCH_SIZE", marker), per_kernel(write_db, "WRITE_SIZE", marker)
    recs = []
    for k in f:
        if prefixes and not any(k.startswith(p) for p in prefixes):
            continue
        fv, wv = f[k], w.get(k, [])
        i = max(range(len(fv)), key=lambda j: fv[j][0])
        fk = sum(v for v, _ in fv) / len(fv)
        dur = sum(d for _, d in fv) / len(fv)
        wk = sum(v for v, _ in wv) / max(len(wv), 1)
        recs.append({"kernel": k, "launches": len(fv), "fetch_size_kb_avg": round(fk, 2), "write_size_kb_avg": round(wk, 2),
                     "hbm_bytes_per_launch": int(round((2 * fk + wk) * 1024)), "avg_duration_us_under_pmc": round(dur / 1e3, 1),
                     "largest_launch": {"fetch_size_kb": round(fv[i][0], 2), "write_size_kb": round(wv[i][0], 2) if i < len(wv) else None,
                                        "hbm_bytes": int(round((2 * fv[i][0] + (wv[i][0] if i < len(wv) else 0.0)) * 1024)),
                                        "duration_us_under_pmc": round(fv[i][1] / 1e3, 1)}})
    recs.sort(key=lambda r: -r["hbm_bytes_per_launch"] * r["launches"])
    json.dump({"command": command, "population": population, "units": units, "source_sha16": source_sha16(), "launches_after_marker": marker, "correction": "FETCH_SIZE x 2 on gfx950 for 16 B/lane coalesced streaming reads (MI355X_MICROARCH.md, HBM section); "
               "WRITE_SIZE as reported; both in KB", "kernels": recs[:16]}, open(out_path, "w"), indent=1)
    for r in recs[:8]:
        print(f"{r['kernel'][:48]:48s} launches={r['launches']:5d} hbm_bytes/launch={r['hbm_bytes_per_launch']:12d} dur_us={r['avg_duration_us_under_pmc']}")


if __name__ == "__main__":
    main()
