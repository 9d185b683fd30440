"""CNN-264k by stretch: launches / waits / host work of one proof in throughput mode (DP_TIMING=1 DP_LAUNCH_NAMES=1), and the executor's
per-body accounting of a batch (DP_RX=1: which bodies carry the time), next to the cohort rate"""
import os, sys, time
sys.path.insert(0, os.getcwd())
import numpy as np
import deep_prove_amd as dpa
dev = dpa.Device(0)
mb = dpa.models.cnn_264k()
ctx = dpa.Context.generate(dev, mb.blob())
pr = dpa.Prover(ctx)
x = mb.input(1000)
for i in range(2):
    t0 = time.perf_counter(); proof, out = pr.prove(x); print(f"single proof {1000 * (time.perf_counter() - t0):.1f} ms", flush=True)
conc = int(sys.argv[1]) if len(sys.argv) > 1 else 256
xs = np.stack([mb.input(3000 + i) for i in range(3 * conc)])
for mode in ("0", "1"):
    os.environ["DP_RX"] = mode
    pr.prove_batch(xs[:conc], conc)
    t0 = time.perf_counter(); pr.prove_batch(xs, conc); dt = time.perf_counter() - t0
    print(f"DP_RX={mode}: {len(xs) / dt:.1f} proofs/s ({conc} in flight)", flush=True)
