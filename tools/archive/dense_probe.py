"""Dense-4M: per-phase host wall time (DP_TIMING), persistent-sumcheck cycle counters (DP_SC_DEBUG) of one proof"""
import os, sys, time
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
os.environ["DP_SC_DEBUG"] = "1"
sys.path.insert(0, os.getcwd())
import numpy as np
import deep_prove_amd as dpa
dev = dpa.Device(0)
mb = dpa.models.dense_4m()
ctx = dpa.Context.generate(dev, mb.blob())
pr = dpa.Prover(ctx)
x = mb.input(1000)
pr.prove(x); pr.prove(x)
os.environ["DP_TIMING"] = "1"
t0 = time.perf_counter(); pr.prove(x); print(f"prove {1000 * (time.perf_counter() - t0):.1f} ms")
