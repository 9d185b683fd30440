"""per-phase host wall time (DP_TIMING) averaged over the proofs of one concurrent batch"""
import os, sys, subprocess, collections, re
if len(sys.argv) > 1 and sys.argv[1] == "child":
    os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
    sys.path.insert(0, os.getcwd())
    import numpy as np
    import deep_prove_amd as dpa
    conc = int(sys.argv[2])
    dev = dpa.Device(0); mb = dpa.models.dense_4m(); ctx = dpa.Context.generate(dev, mb.blob()); pr = dpa.Prover(ctx)
    xs = np.stack([mb.input(3000 + i) for i in range(conc)])
    pr.prove_batch(xs, conc)
    os.environ["DP_TIMING"] = "1"
    pr.prove_batch(xs, conc)
else:
    for conc in sys.argv[1:]:
        out = subprocess.run([sys.executable, __file__, "child", conc], capture_output=True, text=True).stderr
        agg = collections.OrderedDict()
        for line in out.splitlines():
            m = re.match(r"\[dp timing\]\s+(.*?)\s+([0-9.]+) ms", line)
            if m:
                name = re.sub(r"layer \d+ ", "layer ", m.group(1))
                agg.setdefault(name, []).append(float(m.group(2)))
        print(f"--- conc={conc}")
        for line in out.splitlines():
            m = re.match(r"\[dp timing\] sumcheck rounds (\d+): device wait ([0-9.]+) ms, host transcript\+algebra ([0-9.]+) ms", line)
            if m:
                agg.setdefault("sumcheck device wait", []).append(float(m.group(2)))
                agg.setdefault("sumcheck host work", []).append(float(m.group(3)))
        for k, v in agg.items():
            print(f"{k:36s} n={len(v):4d} mean={sum(v)/len(v):8.3f} ms  sum/proof={sum(v)/int(conc):8.3f}")
