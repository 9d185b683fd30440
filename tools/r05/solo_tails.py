"""Two Dense-4M proofs at a time in THROUGHPUT mode (device-side Fiat-Shamir, fused protocol tails, a cohort of one): under rocprofv3 --kernel-trace --stats the
tails' durations ALONE on the chip — what a member costs before it shares its CU with hash kernels.  usage: python tools/r05/solo_tails.py [workload] [n]"""
import os, sys, time
os.environ.setdefault("GPU_MAX_HW_QUEUES", "24")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import deep_prove_amd as dpa
wl = sys.argv[1] if len(sys.argv) > 1 else "dense_4m"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 6
dev = dpa.Device(0)
mb = getattr(dpa.models, wl)()
ctx = dpa.Context.generate(dev, mb.blob())
pr = dpa.Prover(ctx)
xs = np.stack([mb.input(3000 + i) for i in range(n)])
pr.prove(xs[0])
pr.prove_batch(xs[:2], 2)  # (one proof in flight would run in latency mode: two, each alone on its stream — cohorts of one — on an otherwise idle chip)
t0 = time.perf_counter()
for i in range(0, n - 1, 2): pr.prove_batch(xs[i:i + 2], 2)
dt = time.perf_counter() - t0
print(f"{wl}: {n} proofs two at a time in throughput mode: {1000 * dt / (n // 2):.1f} ms per pair", flush=True)
ctx.free()
