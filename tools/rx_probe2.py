"""resident executor diagnosis: RX batch vs sequential proofs — first differing word, number of differing words, verifier verdict,
and whether two RX runs agree with each other (a race differs from run to run, a logic bug does not)"""
import os, sys, time
sys.path.insert(0, os.getcwd())
import numpy as np
import deep_prove_amd as dpa
width = int(sys.argv[1]) if len(sys.argv) > 1 else 64
n = int(sys.argv[2]) if len(sys.argv) > 2 else 4
dev = dpa.Device(0)
mb = dpa.models.mlp(2, width, config=43)
ctx = dpa.Context.generate(dev, mb.blob())
pr = dpa.Prover(ctx)
xs = np.stack([mb.input(5000 + i) for i in range(n)])
os.environ["DP_RX"] = "0"
seq = [pr.prove(x) for x in xs]
cp, co, _ = pr.prove_batch(xs, n)
print("cohort batch == sequential:", all((cp[i] == seq[i][0]).all() for i in range(n)), flush=True)
os.environ["DP_RX"] = "1"
runs = []
for rep in range(2):
    proofs, outs, _ = pr.prove_batch(xs, n)
    runs.append(proofs)
    for i in range(n):
        a, b = proofs[i], seq[i][0]
        if a.size != b.size:
            print(f"run {rep} proof {i}: SIZE {a.size} vs {b.size}"); continue
        d = np.nonzero(a != b)[0]
        verdict = "-"
        try:
            dpa.verify(ctx.verifier_blob(), a, xs[i], outs[i]); verdict = "verifies"
        except Exception as e:  # noqa: BLE001
            verdict = "REJECTED: " + str(e)[:80]
        print(f"run {rep} proof {i}: {d.size} of {a.size} words differ, first at {int(d[0]) if d.size else -1}, last at {int(d[-1]) if d.size else -1}; {verdict}", flush=True)
same = all((runs[0][i].size == runs[1][i].size) and (runs[0][i] == runs[1][i]).all() for i in range(n))
print("two RX runs agree with each other:", same, flush=True)
print("RX_PARITY_OK" if all((runs[0][i] == seq[i][0]).all() for i in range(n)) else "RX_MISMATCH", flush=True)
