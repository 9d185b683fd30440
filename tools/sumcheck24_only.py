"""standalone 2^nv sumcheck (BASELINE config 5 on one GPU; nv = 24, or argv[2]: 26 for the size the sharded estimate is quoted at), a few repetitions — the
command the rocprofv3 PMC passes wrap. usage: sumcheck24_only.py [repetitions] [nv]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))  # (the repository root, wherever the command is started from)
import numpy as np
import deep_prove_amd as dpa
nv, k = (int(sys.argv[2]) if len(sys.argv) > 2 else 24), 3
dev = dpa.Device(0)
n = 1 << nv
tabs = [dpa.Mle.from_base(dev, dpa.models.splitmix64(0xD33B0000 ^ (5 << 32) ^ j, n) % np.uint64(dpa.P)) for j in range(k)]
vp = dpa.VirtualPolynomial(nv)
vp.add_mle_list(tabs)
import hashlib
for _ in range(int(sys.argv[1]) if len(sys.argv) > 1 else 3):
    t0 = time.perf_counter(); proof, finals = dpa.prove_parallel(dev, vp, dpa.Transcript(b"test")); print(f"{1000 * (time.perf_counter() - t0):.3f} ms  proof sha256 {hashlib.sha256(proof.tobytes()).hexdigest()[:16]}")
if os.environ.get("SC24_PROFILE"):
    dev.profile(True); dpa.prove_parallel(dev, vp, dpa.Transcript(b"test")); rep = dev.profile_report(); dev.profile(False)
    for r in sorted(rep, key=lambda r: -r["total_ms"])[:8]:
        print(f'{r["kernel"]:34s} launches {r["launches"]:3d}  total {r["total_ms"]:.4f} ms  {r["alg_bytes"] / max(r["total_ms"], 1e-9) / 1e6:8.1f} GB/s')
