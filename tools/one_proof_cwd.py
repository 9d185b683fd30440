"""three single Dense-4M proofs (the command a rocprofv3 kernel trace wraps to get solo kernel durations); DP_DEVICE_FS=1 makes a
single proof take the throughput path (device-side transcript, fused protocol kernels)"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import deep_prove_amd as dpa
dev = dpa.Device(0)
mb = dpa.models.dense_4m()
ctx = dpa.Context.generate(dev, mb.blob())
pr = dpa.Prover(ctx)
for i in range(3):
    t0 = time.perf_counter(); pr.prove(mb.input(1000 + i)); print("prove wall ms", round(1000 * (time.perf_counter() - t0), 2), flush=True)
