import sys, os, time
sys.path.insert(0, os.getcwd())
import deep_prove_amd as dpa
dev = dpa.Device(0)
mb = dpa.models.dense_4m()
ctx = dpa.Context.generate(dev, mb.blob())
pr = dpa.Prover(ctx)
pr.prove(mb.input(1000))
os.environ["DP_TIMING"] = "1"
t0 = time.perf_counter(); p, o = pr.prove(mb.input(1001)); t1 = time.perf_counter()
print("prove wall ms", 1000 * (t1 - t0), "inner", pr.last_prove_ms, file=sys.stderr)
