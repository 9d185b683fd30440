"""one proof at a time through the THROUGHPUT-mode machinery (prove() with dp_ctx_set_throughput_mode: fused protocol tails, transcript on the device or — DP_HOST_SPONGE=1 — served by the host)
against the latency-mode prove(): is a chain of fused tails shorter than a chain of host-driven persistent sumchecks?  usage: lat_probe.py dense_4m|cnn_264k"""
import hashlib, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import deep_prove_amd as dpa
wl = sys.argv[1] if len(sys.argv) > 1 else "dense_4m"
mb = getattr(dpa.models, wl)()
dev = dpa.Device(0)
ctx = dpa.Context.generate(dev, mb.blob())
pr = dpa.Prover(ctx)
xs = np.stack([mb.input(3000 + i) for i in range(4)])
pr.prove(xs[0])
lat = []
for _ in range(5):
    t0 = time.perf_counter(); p0 = pr.prove(xs[0]); lat.append(1000 * (time.perf_counter() - t0))
dev.set_throughput_mode(True)  # the context's own calls now take the device-side transcript and the fused protocol kernels (no cohort: its own stream)
pr.prove(xs[0])
b1 = []
for _ in range(5):
    t0 = time.perf_counter(); p1 = pr.prove(xs[0]); b1.append(1000 * (time.perf_counter() - t0))
dev.set_throughput_mode(False)
h = lambda p: hashlib.sha256(np.asarray(p[0] if isinstance(p, tuple) else p).tobytes()).hexdigest()[:12]
same = h(p0), h(p1)
print(f"{wl}: prove() {sorted(lat)[2]:.2f} ms (min {min(lat):.2f}); prove() in throughput mode {sorted(b1)[2]:.2f} ms (min {min(b1):.2f}); sha {same}  env: " + " ".join(f"{k}={v}" for k, v in os.environ.items() if k.startswith("DP_")), flush=True)
ctx.free()
