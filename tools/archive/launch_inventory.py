"""stream commands (kernel launches + copies) of ONE Dense-4M proof, by kind"""
import os, sys
sys.path.insert(0, os.getcwd())
import deep_prove_amd as dpa
dev = dpa.Device(0)
mb = dpa.models.dense_4m()
ctx = dpa.Context.generate(dev, mb.blob())
pr = dpa.Prover(ctx)
pr.prove(mb.input(1000))
dev.profile(True)
pr.prove(mb.input(1001))
rep = dev.profile_report()
dev.profile(False)
rep.sort(key=lambda r: -r["launches"])
tot = sum(r["launches"] for r in rep); ms = sum(r["total_ms"] for r in rep)
print(f"total commands {tot}, event-timed ms {ms:.2f}")
for r in rep:
    print(f'{r["launches"]:6d}  {r["total_ms"]:9.3f} ms  {1000 * r["total_ms"] / r["launches"]:9.1f} us avg  {r["kernel"]}')
