"""throughput vs proofs in flight on one GPU (Dense-4M)"""
import os, sys, time
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
sys.path.insert(0, os.getcwd())
import numpy as np
import deep_prove_amd as dpa
dev = dpa.Device(0)
mb = dpa.models.dense_4m()
ctx = dpa.Context.generate(dev, mb.blob())
pr = dpa.Prover(ctx)
print("host cores", os.cpu_count())
for conc in [int(a) for a in sys.argv[1:]] or [1, 2, 4, 8, 16]:
    xs = np.stack([mb.input(3000 + i) for i in range(int(os.environ.get("SWEEP_BATCHES", "2")) * conc)])
    pr.prove_batch(xs[:conc], conc)  # warm (creates the workers)
    t0 = time.perf_counter()
    pr.prove_batch(xs, conc)
    dt = time.perf_counter() - t0
    print(f"conc={conc:3d}  {len(xs) / dt:8.2f} proofs/s   batch of {len(xs)} in {1000 * dt:8.1f} ms", flush=True)
