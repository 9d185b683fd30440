#!/usr/bin/env python
"""Per-kernel register / LDS / scratch use of libdeepprove_hip.so's device code as the compiler reports it
(hipcc -Rpass-analysis=kernel-resource-usage, gfx950) -> CSV. Needs no GPU.  usage: python tools/kernel_resources.py out.csv
(DP_HIPCC_EXTRA="-D..." in the environment: the resources of a variant build)"""
import csv
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
from rocpd_summary import short

FIELDS = ["VGPRs", "AGPRs", "TotalSGPRs", "ScratchSize [bytes/lane]", "Occupancy [waves/SIMD]", "SGPRs Spill", "VGPRs Spill", "LDS Size [bytes/block]"]


def main():
    src = os.path.join(ROOT, "deep-prove_amd", "csrc", "hip_dev.hip")
    with tempfile.TemporaryDirectory() as d:
        r = subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-c", "-Wno-unused-value",
                            "-Rpass-analysis=kernel-resource-usage", "-o", os.path.join(d, "x.o"), src] + os.environ.get("DP_HIPCC_EXTRA", "").split(), capture_output=True, text=True)
    recs, cur = [], None
    for ln in r.stderr.split("\n"):
        m = re.search(r"remark: Function Name: (\S+)", ln)
        if m:
            # the launch form of a body: kg / kc<Body, MAXT, FLAGS> — MAXT = __launch_bounds__, FLAGS 1 = KF_CLAIM (latency mode, 1024 threads), 2 = KF_PRIO
            # (throughput mode, <= 256 threads)
            f = re.search(r"EELi(\d+)ELi(\d+)E", m.group(1))
            cur = {"Kernel": short(m.group(1)), "MAXT": f.group(1) if f else "", "FLAGS": {"0": "none", "1": "claim", "2": "prio"}.get(f.group(2), f.group(2)) if f else ""}
            recs.append(cur)
            continue
        m = re.search(r"remark:\s+([A-Za-z][A-Za-z \[\]/]+): (\d+)", ln)
        if m and cur is not None and m.group(1).strip() in FIELDS:
            cur[m.group(1).strip()] = int(m.group(2))
    recs.sort(key=lambda x: x["Kernel"].split(":", 1)[-1] + x["Kernel"][:2])
    with open(sys.argv[1], "w", newline="") as f:
        w = csv.DictWriter(f, fieldnames=["Kernel", "MAXT", "FLAGS"] + FIELDS)
        w.writeheader()
        for x in recs:
            w.writerow({k: x.get(k, "") for k in ["Kernel", "MAXT", "FLAGS"] + FIELDS})
    spill = [f'{x["Kernel"]}[{x["MAXT"]},{x["FLAGS"]}]' for x in recs if x.get("VGPRs Spill", 0)]
    print(f"{len(recs)} kernels; with VGPR spills: {', '.join(spill) if spill else 'none'}")


if __name__ == "__main__":
    main()
