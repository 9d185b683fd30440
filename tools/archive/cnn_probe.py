"""CNN-264k on the device: single-proof latency, kernel inventory and throughput with several proofs in flight"""
import os, sys, time
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
sys.path.insert(0, os.getcwd())
import numpy as np
import deep_prove_amd as dpa
dev = dpa.Device(0)
mb = dpa.models.cnn_264k()
t0 = time.time(); ctx = dpa.Context.generate(dev, mb.blob()); print(f"setup {time.time() - t0:.2f} s")
pr = dpa.Prover(ctx)
x = mb.input(1000)
for i in range(3):
    t0 = time.perf_counter(); proof, out = pr.prove(x); print(f"prove {1000 * (time.perf_counter() - t0):.1f} ms, {proof.size} words")
dpa.verify(ctx.verifier_blob(), proof, x, out)
dev.profile(True); pr.prove(x); rep = dev.profile_report(); dev.profile(False)
rep.sort(key=lambda r: -r["total_ms"])
tot = sum(r["launches"] for r in rep); ms = sum(r["total_ms"] for r in rep)
print(f"total commands {tot}, event-timed ms {ms:.2f}")
for r in rep[:25]:
    print(f'{r["launches"]:6d}  {r["total_ms"]:9.3f} ms  {1000 * r["total_ms"] / r["launches"]:9.1f} us avg  {r["kernel"]}')
os.environ["DP_TIMING"] = "1"; pr.prove(x); os.environ["DP_TIMING"] = "0"
for conc in (4, 16):
    xs = np.stack([mb.input(3000 + i) for i in range(2 * conc)])
    pr.prove_batch(xs[:conc], conc)
    t0 = time.perf_counter(); pr.prove_batch(xs, conc); dt = time.perf_counter() - t0
    print(f"conc={conc:3d}  {len(xs) / dt:8.2f} proofs/s", flush=True)
