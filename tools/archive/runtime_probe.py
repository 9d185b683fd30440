"""does the HIP runtime that ends up loaded matter? modes: none (no torch), torch_first, lib_first. Prints the loaded libamdhip64,
the Poseidon2 probe rate and a 3-wave Dense-4M batch rate."""
import os, sys, time
mode = sys.argv[1]
sys.path.insert(0, os.getcwd())
os.environ.setdefault("GPU_MAX_HW_QUEUES", "24")
if mode == "torch_first":
    import torch
    torch.cuda.is_available()
import numpy as np
import deep_prove_amd as dpa
dpa._lib.load()
if mode == "lib_first":
    import torch
    torch.cuda.is_available()
hip = sorted(set(l.split()[-1] for l in open("/proc/self/maps") if "libamdhip64" in l))
dev = dpa.Device(0)
r1 = dev.probe_compress_rate(1 << 21, 5)
mb = dpa.models.dense_4m(); ctx = dpa.Context.generate(dev, mb.blob()); pr = dpa.Prover(ctx)
xs = np.stack([mb.input(3000 + i) for i in range(3 * 256)])
pr.prove_batch(xs[:256], 256)
t0 = time.perf_counter(); pr.prove_batch(xs, 256); dt = time.perf_counter() - t0
r2 = dev.probe_compress_rate(1 << 21, 5)
if mode != "none":
    torch.cuda.synchronize()
print(f"{mode:12s} hip={hip} probe before {r1 / 1e9:.3f} after {r2 / 1e9:.3f} Gcompress/s, batch {len(xs) / dt:.1f} proofs/s", flush=True)
