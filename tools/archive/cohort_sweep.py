"""throughput vs (proofs in flight, cohort size) on one GPU: python tools/cohort_sweep.py [workload] conc:cohort[:threads] ...
cohort 0 = every proof on its own stream. DP_TIMING=1 adds the host-work accounting of the first contexts."""
import os, sys, time
os.environ.setdefault("GPU_MAX_HW_QUEUES", "24")
sys.path.insert(0, os.getcwd())
import numpy as np
import deep_prove_amd as dpa
args = sys.argv[1:]
wl = "dense_4m"
if args and ":" not in args[0]:
    wl = args.pop(0)
dev = dpa.Device(0)
mb = getattr(dpa.models, wl)()
ctx = dpa.Context.generate(dev, mb.blob())
pr = dpa.Prover(ctx)
pr.prove(mb.input(2999))  # the footprint of one proof sizes the arenas of the batch workers
print("host cores", os.cpu_count(), flush=True)
for spec in args or ["8:8", "8:0", "24:8", "24:0"]:
    f = [int(v) for v in spec.split(":")]
    conc, co = f[0], f[1]
    if len(f) > 2:
        os.environ["DP_HOST_THREADS"] = str(f[2])
    else:
        os.environ.pop("DP_HOST_THREADS", None)
    os.environ["DP_COHORT"] = str(co)
    xs = np.stack([mb.input(3000 + i) for i in range(int(os.environ.get("SWEEP_BATCHES", "2")) * conc)])
    pr.prove_batch(xs[:conc], conc)  # warm (creates the workers)
    print(f"--- conc={conc} cohort={co}", file=sys.stderr, flush=True)
    t0 = time.perf_counter()
    pr.prove_batch(xs, conc)
    dt = time.perf_counter() - t0
    print(f"conc={conc:3d} in_flight={pr.in_flight():3d} cohort={co:2d} threads={f[2] if len(f) > 2 else 'auto'}  {len(xs) / dt:8.2f} proofs/s   batch of {len(xs)} in {1000 * dt:8.1f} ms", flush=True)
