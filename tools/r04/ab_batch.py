"""A/B probe (torch-free): <workload> proofs in throughput mode on whichever library DP_LIB_VARIANT names; prints proofs/s, the single-proof latency and the
sha256 of one throughput-mode proof (identical across builds when nothing but scheduling changed).
usage: python tools/r04/ab_batch.py dense_4m|cnn_264k|transformer_layer <in flight> <batches>"""
import hashlib, os, sys, time
os.environ.setdefault("GPU_MAX_HW_QUEUES", "24")
_root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, _root)
import numpy as np
import deep_prove_amd as dpa
wl = sys.argv[1]; conc = int(sys.argv[2]); nb = int(sys.argv[3]) if len(sys.argv) > 3 else 3
mb = dpa.models.transformer_layer(64, 256, 4, 64, 1024, config=66) if wl == "transformer_layer" else getattr(dpa.models, wl)()
dev = dpa.Device(0)
ctx = dpa.Context.generate(dev, mb.blob())
pr = dpa.Prover(ctx)
xs = np.stack([mb.input(3000 + i) for i in range(conc * nb)])
pr.prove(xs[0])
lat = []
for _ in range(3):
    t0 = time.perf_counter(); pr.prove(xs[0]); lat.append(1000 * (time.perf_counter() - t0))
pr.prove_batch(xs[:conc], conc)
t0 = time.perf_counter(); proofs, outs, _ = pr.prove_batch(xs, conc); dt = time.perf_counter() - t0
sel = list(range(8)) + list(range(len(xs) - 8, len(xs)))  # the first wave (prepared by the workers themselves) and the last (prepared ahead by the helper threads)
v, _ = dpa.verify_batch(ctx.verifier_blob(), [proofs[i] for i in sel], xs[sel], [outs[i] for i in sel], dev=dev)
print(f"{wl} variant={os.environ.get('DP_LIB_VARIANT', 'release')}: {len(xs) / dt:.1f} proofs/s ({pr.in_flight()} in flight, {len(xs)} proofs, {dt * 1000:.0f} ms); single proof {sorted(lat)[1]:.1f} ms; "
      f"rejected of 16: {int(v.sum())}; sha256(proof 5) {hashlib.sha256(proofs[5].tobytes()).hexdigest()[:16]} sha256(proof -3) {hashlib.sha256(proofs[-3].tobytes()).hexdigest()[:16]}", flush=True)
ctx.free()
