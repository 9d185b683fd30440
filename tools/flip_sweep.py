"""Exhaustive single-bit malleability sweep of the host verifier: flip bit `BIT` (default 0) of every word of a golden proof
(tests/golden/*.npz: verifier_blob, proof, input, output) in [lo, hi) and report the words whose flip still verifies.
A bound proof stream has none. Host only (dp_verify), ~100 words/s.
  python tools/flip_sweep.py mlp_w8.npz [lo hi]      BIT=63 python tools/flip_sweep.py cnn_tiny.npz 0 20000
tests/support/fuzz_proof.py is the random-mutation counterpart that runs in the CPU suite."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
import deep_prove_amd as dpa

def main():
    name = sys.argv[1]
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden", name))
    p0 = g["proof"]
    lo = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    hi = min(int(sys.argv[3]) if len(sys.argv) > 3 else p0.size, p0.size)
    bit = np.uint64(1) << np.uint64(int(os.environ.get("BIT", "0")))
    dpa.verify(g["verifier_blob"], p0, g["input"], g["output"])  # the golden proof itself must verify
    acc, t0 = [], time.time()
    for i in range(lo, hi):
        p = p0.copy(); p[i] ^= bit
        try:
            dpa.verify(g["verifier_blob"], p, g["input"], g["output"]); acc.append(i)
        except dpa.DeepProveError:
            pass
    print("%s: words [%d, %d) of %d, bit %s: %d accepted flips %s (%.0f s)" % (name, lo, hi, p0.size, os.environ.get("BIT", "0"), len(acc), acc[:40], time.time() - t0))
    return 1 if acc else 0

if __name__ == "__main__":
    sys.exit(main())
