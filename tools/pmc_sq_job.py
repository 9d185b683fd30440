#!/usr/bin/env python
"""Chip-level VALU accounting of the timed job from a rocprofv3 --pmc pass (SQ_INSTS_VALU SQ_WAVES GRBM_GUI_ACTIVE SQ_BUSY_CYCLES) over
`tools/profile_batch.py dense_4m <in flight>`: counter collection serialises the dispatches, so durations mean nothing here — the INSTRUCTION COUNTS do.
Only the cohort launches count (kc<Body>: the two batches of the command; the latency-mode proof and Context::generate run as kg<Body>), divided by the
proofs they prove. With the rate of the un-profiled job this gives what fraction of the chip's VALU issue slots the job uses:
  valu_issue_util = wave_instr_per_proof x proofs_per_s x CYCLES_PER_WAVE_INSTR / (1024 SIMDs x clock)
CYCLES_PER_WAVE_INSTR = 4 (a wave64 instruction holds its 16-lane SIMD for 4 cycles; profiles/r02_instr_rate_gfx950.txt measures 4.2-4.5 for the 64-bit
integer ops of Goldilocks arithmetic and 2.6 for v_add_u32: 4 is the documented issue cost, the measured mix is within 15 % of it).
usage: python tools/pmc_sq_job.py <results.db> <proofs> <proofs_per_s of the un-profiled job> <out.json> "<command>" [<probe results.db> <nodes per launch>]"""
import json
import sqlite3
import sys

sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.abspath(__file__)))
from rocpd_summary import short
from srchash import source_sha16

SIMDS, CLOCK_HZ, CYC = 1024, 2.4e9, 4.0


def counters(db_path, want_prefix):
    db = sqlite3.connect(db_path)
    acc = {}
    for name, counter, value, did in db.execute("select kernel_name, counter_name, value, dispatch_id from counters_collection order by id"):
        k = short(name)
        if not k.startswith(want_prefix):
            continue
        a = acc.setdefault(k, {"d": set(), "c": {}})
        a["d"].add(did)
        a["c"][counter] = a["c"].get(counter, 0.0) + float(value)
    return acc


def main():
    db_path, proofs, rate, out_path, command = sys.argv[1], int(sys.argv[2]), float(sys.argv[3]), sys.argv[4], sys.argv[5]
    acc = counters(db_path, "kc:")
    rows = []
    for k, a in acc.items():
        v = a["c"].get("SQ_INSTS_VALU", 0.0)
        rows.append({"kernel": k[3:], "cohort_launches": len(a["d"]), "valu_wave_instr_per_proof": round(v / proofs, 1), "waves_per_proof": round(a["c"].get("SQ_WAVES", 0.0) / proofs, 2)})
    rows.sort(key=lambda r: -r["valu_wave_instr_per_proof"])
    total = sum(r["valu_wave_instr_per_proof"] for r in rows)
    for r in rows:
        r["share"] = round(r["valu_wave_instr_per_proof"] / total, 4) if total else 0.0
    cap = SIMDS * CLOCK_HZ / CYC
    doc = {"command": command, "population": "dense_4m_throughput_mode_cohort_launches", "source_sha16": source_sha16(), "proofs": proofs, "proofs_per_s_unprofiled": rate,
           "valu_wave_instr_per_proof": round(total, 1), "valu_wave_instr_per_s": round(total * rate, 1),
           "chip_issue_capacity_wave_instr_per_s": cap, "assumed": {"simds": SIMDS, "clock_hz": CLOCK_HZ, "cycles_per_wave_instr": CYC},
           "valu_issue_util": round(total * rate / cap, 4),
           "one_wave_share": round(sum(r["valu_wave_instr_per_proof"] for r in rows if "_tail" in r["kernel"] or "persist" in r["kernel"] or "sc_small" in r["kernel"]) / total, 4) if total else None,
           "hash_share": round(sum(r["valu_wave_instr_per_proof"] for r in rows if "merkle" in r["kernel"]) / total, 4) if total else None,
           "note": "instruction counts from a counter pass (dispatches serialised: durations not comparable); one-wave kernels occupy an issue port per wave they run on, "
                   "so their share of the issue SLOTS they deny to others is larger than their instruction share", "kernels": rows[:24]}
    if len(sys.argv) > 7:  # the compress probe: k_merkle_layer on a chip-filling layer (kg form)
        pa = counters(sys.argv[6], "kg:k_merkle_layer")
        nodes = float(sys.argv[7])
        best = None
        for k, a in pa.items():
            n = len(a["d"])
            per = a["c"].get("SQ_INSTS_VALU", 0.0) / n / (nodes / 64.0)  # wave instructions per wave of 64 nodes = VALU instructions per compress
            if k == "kg:k_merkle_layer":
                best = per
        if best:
            bound = cap * 64.0 / best  # compress()/s if every issue slot of the chip went to this kernel at CYC cycles per wave instruction
            doc["compress"] = {"valu_instr_per_compress": round(best, 1), "peak_valu_bound_compress_per_s": round(bound, 1),
                               "note": "dynamic VALU instructions of one Poseidon2 compress (2 permutations, ~1040 multiplications) from SQ_INSTS_VALU over the probe's launches; "
                                       "the bound prices every one of them at 4 cycles on 1024 SIMDs at 2.4 GHz"}
    json.dump(doc, open(out_path, "w"), indent=1)
    print(json.dumps({k: v for k, v in doc.items() if k not in ("kernels", "note", "assumed")}))
    for r in rows[:12]:
        print(r)


if __name__ == "__main__":
    main()
