"""Dense-4M (or argv[1]) proofs in flight, torch-free — the command the rocprofv3 --kernel-trace --stats pass of the cohort
scheme wraps: one single proof (latency mode), one warm batch that creates the workers, one measured batch.
usage: python tools/profile_batch.py [workload] [in_flight] [waves]   (waves: proofs per batch = waves x in_flight, both batches; default 1)"""
import os, sys, time
os.environ.setdefault("GPU_MAX_HW_QUEUES", "24")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import deep_prove_amd as dpa
wl = sys.argv[1] if len(sys.argv) > 1 else "dense_4m"
conc = int(sys.argv[2]) if len(sys.argv) > 2 else 192
dev = dpa.Device(0)
mb = dpa.models.transformer_layer(64, 256, 4, 64, 1024, config=66) if wl == "transformer_layer" else getattr(dpa.models, wl)()  # (golden case 14: the size bench.py times)
ctx = dpa.Context.generate(dev, mb.blob())
pr = dpa.Prover(ctx)
waves = int(sys.argv[3]) if len(sys.argv) > 3 else 1
xs = np.stack([mb.input(3000 + i) for i in range(conc * waves)])
pr.prove(xs[0])
pr.prove_batch(xs, conc)
t0 = time.perf_counter()
pr.prove_batch(xs, conc)
dt = time.perf_counter() - t0
print(f"{wl}: {len(xs)} proofs, {pr.in_flight()} in flight: {len(xs) / dt:.1f} proofs/s under the profiler", flush=True)
ctx.free()
