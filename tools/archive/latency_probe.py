"""single-proof latency, several proofs in a row (latency mode), with and without keeping the proofs alive"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))  # (the repository root, wherever the command is started from)
import deep_prove_amd as dpa
dev = dpa.Device(0)
mb = getattr(dpa.models, sys.argv[1] if len(sys.argv) > 1 else 'dense_4m')()
ctx = dpa.Context.generate(dev, mb.blob())
pr = dpa.Prover(ctx)
keep = []
for i in range(6):
    t0 = time.perf_counter(); p, o = pr.prove(mb.input(1000 + (i % 2))); t1 = time.perf_counter()
    print(f"proof {i}: wall {1000 * (t1 - t0):.2f} ms, inside the library {pr.last_prove_ms:.2f} ms", file=sys.stderr)
    if i >= 3:
        keep.append(p)
    del p
