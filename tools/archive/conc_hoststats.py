"""host-side launch cost / waits per device context with several proofs in flight (DP_TIMING)"""
import os, sys, time
os.environ.setdefault("GPU_MAX_HW_QUEUES", "24")
os.environ.setdefault("DP_TIMING", "1")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))  # (the repository root, wherever the command is started from)
import numpy as np
import deep_prove_amd as dpa
conc = int(sys.argv[1])
dev = dpa.Device(0); mb = getattr(dpa.models, sys.argv[2] if len(sys.argv) > 2 else 'dense_4m')(); ctx = dpa.Context.generate(dev, mb.blob()); pr = dpa.Prover(ctx)
xs = np.stack([mb.input(3000 + i) for i in range(4 * conc)])
pr.prove(xs[0])  # the arena of a worker follows the footprint of a proof already proved
pr.prove_batch(xs[:conc], conc)
c0 = time.process_time(); t0 = time.perf_counter(); pr.prove_batch(xs, conc); dt = time.perf_counter() - t0; cpu = time.process_time() - c0
print(f"conc={conc} {len(xs)/dt:.1f} proofs/s; process CPU {cpu:.2f} s over {dt:.2f} s wall = {cpu/dt:.1f} cores busy (polling included), {1000*cpu/len(xs):.2f} CPU-ms per proof", file=sys.stderr)
