"""Exhaustive single-bit malleability sweep of the host verifier: flip bit `BIT` (default 0) of every word of a golden proof
(tests/golden/*.npz: verifier_blob, proof, input, output) in [lo, hi) and report the words whose flip still verifies.
A bound proof stream has none. Host only (dp_verify), ~30-100 words/s per process; JOBS processes (default: the machine's cores) share the range.
  python tools/flip_sweep.py mlp_w8.npz [lo hi]      BIT=63 JOBS=4 python tools/flip_sweep.py cnn_tiny.npz 0 20000
tests/support/fuzz_proof.py is the random-mutation counterpart that runs in the CPU suite; profiles/r02_flip_sweep.txt holds the results."""
import multiprocessing as mp
import os
import sys
import time
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
import numpy as np

os.environ.setdefault("DP_VERIFY_THREADS", "1")  # one verification per process: the processes are the parallelism


def sweep(job):
    name, lo, hi, bit = job
    import deep_prove_amd as dpa
    g = np.load(os.path.join(ROOT, "tests", "golden", name))
    p0, acc = g["proof"], []
    for i in range(lo, hi):
        p = p0.copy(); p[i] ^= np.uint64(1) << np.uint64(bit)
        try:
            dpa.verify(g["verifier_blob"], p, g["input"], g["output"]); acc.append(i)
        except dpa.DeepProveError:
            pass
    return acc


def main():
    name = sys.argv[1]
    import deep_prove_amd as dpa
    g = np.load(os.path.join(ROOT, "tests", "golden", name))
    n = g["proof"].size
    lo = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    hi = min(int(sys.argv[3]) if len(sys.argv) > 3 else n, n)
    bit = int(os.environ.get("BIT", "0"))
    jobs = max(1, int(os.environ.get("JOBS", str(os.cpu_count() or 1))))
    dpa.verify(g["verifier_blob"], g["proof"], g["input"], g["output"])  # the golden proof itself must verify
    step = max(1, (hi - lo + 4 * jobs - 1) // (4 * jobs))
    parts = [(name, a, min(a + step, hi), bit) for a in range(lo, hi, step)]
    t0 = time.time()
    with mp.get_context("spawn").Pool(jobs) as pool:
        acc = sorted(i for part in pool.map(sweep, parts) for i in part)
    print("%s: words [%d, %d) of %d, bit %d: %d accepted flips %s (%.0f s, %d processes)" % (name, lo, hi, n, bit, len(acc), acc[:40], time.time() - t0, jobs))
    return 1 if acc else 0


if __name__ == "__main__":
    sys.exit(main())
