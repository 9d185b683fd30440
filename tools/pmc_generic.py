#!/usr/bin/env python
"""Per-kernel means of arbitrary rocprofv3 --pmc counters (one rocpd database) -> JSON + table. Used for the SQ pass that backs
the "VALU-integer bound" statements: SQ_INSTS_VALU (wave instructions), SQ_ACTIVE_INST_VALU (quad-cycles the VALU was busy),
SQ_WAVE_CYCLES, SQ_BUSY_CYCLES, GRBM_GUI_ACTIVE. ROCm 7.2 ships no gfx950 derived counters (MI355X_MICROARCH.md, PMC slots), so:
  valu_issue_util = 4 * SQ_INSTS_VALU / (1024 SIMDs * GRBM_GUI_ACTIVE / 8)
a wave64 VALU instruction occupies its 16-lane SIMD for 4 cycles (tools/instr_rate.hip: 4.2-4.5 for the 64-bit integer ops of
Goldilocks arithmetic), SQ_INSTS_VALU counts wave instructions of the whole chip (SQ_WAVES per launch = grid x waves per block
confirms that all 8 XCDs are counted), and GRBM_GUI_ACTIVE comes back summed over the 8 XCDs (GRBM / 8 / duration = 2.1-2.2 GHz,
the clock under counter collection). On gfx950 SQ_ACTIVE_INST_VALU equals SQ_INSTS_VALU (it counts issues, not quad-cycles).
usage: python tools/pmc_generic.py <x_results.db> <out.json> "<command>" [kernel-prefix ...]"""
import json
import sqlite3
import sys

sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.abspath(__file__)))
from rocpd_summary import short as _short


def short(name):
    s = _short(name)
    return s.split(":", 1)[1] if s[:3] in ("kg:", "kc:") else s


def main():
    db_path, out_path, command = sys.argv[1:4]
    prefixes = sys.argv[4:]
    db = sqlite3.connect(db_path)
    acc = {}
    for name, counter, value, dur, did in db.execute("select kernel_name, counter_name, value, duration, dispatch_id from counters_collection order by id"):
        k = short(name)
        if prefixes and not any(k.startswith(p) for p in prefixes):
            continue
        a = acc.setdefault(k, {"dispatches": set(), "dur": {}, "c": {}})
        a["dispatches"].add(did); a["dur"][did] = float(dur)
        a["c"][counter] = a["c"].get(counter, 0.0) + float(value)
    recs = []
    for k, a in acc.items():
        n = len(a["dispatches"])
        r = {"kernel": k, "launches": n, "avg_duration_us_under_pmc": round(sum(a["dur"].values()) / n / 1e3, 2)}
        for c, v in sorted(a["c"].items()):
            r[c + "_per_launch"] = round(v / n, 1)
        if "SQ_INSTS_VALU" in a["c"] and a["c"].get("GRBM_GUI_ACTIVE"):
            r["valu_issue_util"] = round(4.0 * a["c"]["SQ_INSTS_VALU"] / (1024.0 * a["c"]["GRBM_GUI_ACTIVE"] / 8.0), 4)
            r["effective_clock_ghz"] = round(a["c"]["GRBM_GUI_ACTIVE"] / 8.0 / sum(a["dur"].values()), 3)
        recs.append(r)
    recs.sort(key=lambda r: -r["avg_duration_us_under_pmc"] * r["launches"])
    json.dump({"command": command, "kernels": recs[:24]}, open(out_path, "w"), indent=1)
    for r in recs[:10]:
        print({k: v for k, v in r.items()})


if __name__ == "__main__":
    main()
