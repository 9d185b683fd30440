"""persistent-sumcheck cycle counters (DP_SC_DEBUG) with several proofs in flight"""
import os, sys
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
os.environ["DP_SC_DEBUG"] = "1"
sys.path.insert(0, os.getcwd())
import numpy as np
import deep_prove_amd as dpa
conc = int(sys.argv[1])
dev = dpa.Device(0); mb = dpa.models.dense_4m(); ctx = dpa.Context.generate(dev, mb.blob()); pr = dpa.Prover(ctx)
xs = np.stack([mb.input(3000 + i) for i in range(conc)])
pr.prove_batch(xs, conc)
pr.prove_batch(xs, conc)
