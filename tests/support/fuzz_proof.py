"""dp_verify on mutated proof streams (TEST INFRASTRUCTURE): flipped bits, huge / zero lengths, truncations, spliced ranges. Every call must
return — accepted (only when the mutation undid itself) or DP_ERR_* — and never crash or hang."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import deep_prove_amd as dpa
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 1)
gs = [np.load(os.path.join(root, "golden", n)) for n in ("mlp_w8.npz", "cnn_tiny.npz", "seq_mlp.npz")]
acc = rej = forged = 0
for it in range(int(sys.argv[2]) if len(sys.argv) > 2 else 900):
    g = gs[it % len(gs)]
    p = g["proof"].copy()
    for _ in range(int(rng.integers(1, 4))):
        mode = int(rng.integers(0, 6)); pos = int(rng.integers(0, p.size))
        if mode == 0: p[pos] ^= np.uint64(1) << np.uint64(rng.integers(0, 64))
        elif mode == 1: p[pos] = np.uint64(rng.integers(0, 2**63))
        elif mode == 2: p[pos] = np.uint64([0, 1, 2**32, 2**62, 2**64 - 1, 0xFFFFFFFF00000001][int(rng.integers(0, 6))])
        elif mode == 3: p = p[:max(1, int(rng.integers(1, p.size)))]
        elif mode == 4:
            a, b = sorted(int(v) for v in rng.integers(0, p.size, size=2)); p = np.concatenate([p[:a], p[b:]])
        else:
            a = int(rng.integers(0, p.size)); p = np.concatenate([p[:a], p[a:a + 50], p[a:]])
        if p.size == 0: p = np.zeros(1, dtype=np.uint64)
    try:
        dpa.verify(g["verifier_blob"], p, g["input"], g["output"]); acc += 1
        if p.size != g["proof"].size or (p != g["proof"]).any():
            forged += 1
            print("ACCEPTED a stream that differs from the proof: iteration", it)
    except dpa.DeepProveError:
        rej += 1
print("fuzz done: accepted", acc, "rejected", rej, "accepted-but-different", forged)
sys.exit(1 if forged else 0)
