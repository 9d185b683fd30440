#!/usr/bin/env python
"""sha256 (first 16 hex digits) over the device / host sources of libdeepprove_hip.so (deep-prove_amd/csrc/*, sorted by name): every counter pass and diagnostic
timing a profile file under profiles/ quotes carries it (`source_sha16`), and bench.py quotes such a figure only when it was collected on the sources it runs —
round 4's config-5 `traffic` came from a pass older than the kernel it was attached to.  usage: python tools/srchash.py"""
import hashlib
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def source_sha16():
    d = os.path.join(ROOT, "deep-prove_amd", "csrc")
    h = hashlib.sha256()
    for name in sorted(os.listdir(d)):
        h.update(name.encode())
        h.update(open(os.path.join(d, name), "rb").read())
    return h.hexdigest()[:16]


if __name__ == "__main__":
    print(source_sha16())
