#!/usr/bin/env python
"""The one-workgroup protocol tails against their own floor -> the JSON bench.py quotes as `tail_roofline`.
A member of a merged k_logup_tail launch is a chain of Poseidon2 permutations on ONE wave (the transcript's sponge): nothing in the kernel can be faster than
    floor = permutations per member x the permutation's cost in a loop on an otherwise idle wave (tools/r04/p2l_bench.hip, profiles/r04_p2l_bench_limb_sponge.txt).
Inputs: the stderr of two diagnostic runs of `tools/archive/conc_hoststats.py 448` (Dense-4M, 448 proofs in flight, lock-step cohorts of ~20):
  <wgtimes.txt>  library built with -DDP_WG_TIMES: entry -> exit of every member of every merged launch (two clock reads per member);
  <phases.txt>   library built with -DDP_WG_TIMES -DDP_WG_PHASES: counts the permutations (its clock reads around every permutation inflate the times: only the COUNT is used).
usage: python tools/tail_roofline.py <wgtimes.txt> <phases.txt> <out.json> [perm_us]"""
import json
import os
import re
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from srchash import source_sha16


def main():
    wg, ph, out = sys.argv[1], sys.argv[2], sys.argv[3]
    perm_us = float(sys.argv[4]) if len(sys.argv) > 4 else 7.3
    last = [ln for ln in open(wg) if "k_logup_tail:" in ln and "merged launches" in ln][-1]  # (the last dump covers the measured batch)
    m = re.search(r"(\d+) merged launches, ([\d.]+) members each: first entry -> last exit (\d+) us; median member (\d+) us; slowest - fastest member (\d+) us; "
                  r"start skew \(last entry - first entry\) (\d+) us; last exit - first exit (\d+) us", last)
    launches, members, span, median, spread, skew, exit_spread = int(m.group(1)), float(m.group(2)), *[float(m.group(i)) for i in range(3, 8)]
    rate = [ln for ln in open(wg) if "proofs/s" in ln][-1]
    perms = [ln for ln in open(ph) if "permutations," in ln][-1]
    nperm = int(re.search(r"\((\d+) permutations", perms).group(1))
    nmemb, group = 0, 0
    for ln in open(ph):  # the member counts (one line per column-length class) printed right before the LAST permutation count
        if "sponge-free time" in ln:
            group += int(re.search(r"; (\d+) members", ln).group(1))
        elif "permutations," in ln:
            nmemb, group = group, 0
    if not nmemb:
        raise SystemExit("tail_roofline: no member count in the phases log")
    per_member = nperm / nmemb
    floor_us = per_member * perm_us
    doc = {"kernel": "kc:k_logup_tail", "population": "dense_4m_in_flight_cohort_launches", "source_sha16": source_sha16(),
           "bound": "latency of one wave: the transcript's sponge is a dependent chain of Poseidon2 permutations",
           "permutations_per_member": round(per_member, 1), "permutation_us_in_a_loop": perm_us, "floor_us_per_member": round(floor_us, 1),
           "median_member_us": median, "merged_launch_us": span, "merged_launches": launches, "members_per_launch": members,
           "start_skew_us": skew, "slowest_minus_fastest_member_us": spread, "last_exit_minus_first_exit_us": exit_spread,
           "frac_member": round(floor_us / median, 4), "frac_merged_launch": round(floor_us / span, 4),
           "job_line": rate.strip()[:160],
           "note": "frac_member = floor / median member, frac_merged_launch = floor / (first entry -> last exit of a merged launch): what the lock step of a cohort pays on top. "
                   "Round 4 (profiles/r04_wgtimes_448_limb_sponge.txt): median member 3864 us, merged launch 5248-6478 us = 0.19-0.25 of the floor.",
           "inputs": [os.path.relpath(wg), os.path.relpath(ph)]}
    json.dump(doc, open(out, "w"), indent=1)
    print(json.dumps({k: doc[k] for k in ("permutations_per_member", "floor_us_per_member", "median_member_us", "merged_launch_us", "frac_member", "frac_merged_launch")}))


if __name__ == "__main__":
    main()
