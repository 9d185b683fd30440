"""Dense-4M (or the workload named in argv[1]) latency-mode proofs for the rocprofv3 counter passes, with the launches of
Context::generate and of the warm-up proof SEPARATED from the measured proofs: a dp_verify_batch call (kernel k_merkle_paths, used
by nothing else) sits between them as a marker, and tools/pmc_summary.py --after-marker k_merkle_paths drops every launch up to it.
The population of the summary is then exactly `argv[2]` proofs (default 3) — what bench.py's kernel_profile times."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import deep_prove_amd as dpa
workload = sys.argv[1] if len(sys.argv) > 1 else "dense_4m"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 3
dev = dpa.Device(0)
mb = {"dense_4m": dpa.models.dense_4m, "cnn_264k": dpa.models.cnn_264k}[workload]()
ctx = dpa.Context.generate(dev, mb.blob())
pr = dpa.Prover(ctx)
x = mb.input(1000)
proof, out = pr.prove(x)  # warm-up
verdicts, _ = dpa.verify_batch(ctx.verifier_blob(), [proof], np.stack([x]), [out], dev=dev)  # the marker launch (k_merkle_paths)
assert not verdicts.any()
for i in range(n):
    t0 = time.perf_counter(); pr.prove(mb.input(1001 + i)); print("prove wall ms", round(1000 * (time.perf_counter() - t0), 2), flush=True)
