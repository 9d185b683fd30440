"""is the process CPU-throttled by its cgroup while proving? (cpu.stat nr_throttled / throttled_usec around a batch)"""
import os, sys, time
os.environ.setdefault("GPU_MAX_HW_QUEUES", "24")
sys.path.insert(0, os.getcwd())
import numpy as np
import deep_prove_amd as dpa
def stat():
    d = {}
    try:
        for line in open("/sys/fs/cgroup/cpu.stat"):
            k, v = line.split(); d[k] = int(v)
    except OSError:
        pass
    return d
conc = int(sys.argv[1])
dev = dpa.Device(0); mb = dpa.models.dense_4m(); ctx = dpa.Context.generate(dev, mb.blob()); pr = dpa.Prover(ctx)
xs = np.stack([mb.input(3000 + i) for i in range(3 * conc)])
pr.prove_batch(xs[:conc], conc)
s0, c0, t0 = stat(), os.times(), time.perf_counter()
pr.prove_batch(xs, conc)
dt = time.perf_counter() - t0; s1, c1 = stat(), os.times()
print(f"threads={os.environ.get('DP_HOST_THREADS')} conc={conc}: {len(xs)/dt:.1f} proofs/s; process cpu {(c1.user - c0.user + c1.system - c0.system) / dt:.1f} cores (user {(c1.user-c0.user)/dt:.1f}, sys {(c1.system-c0.system)/dt:.1f}); "
      f"cgroup usage {(s1.get('usage_usec',0)-s0.get('usage_usec',0))/1e6/dt:.1f} cores, throttled periods {s1.get('nr_throttled',0)-s0.get('nr_throttled',0)} of {s1.get('nr_periods',0)-s0.get('nr_periods',0)}, throttled time {(s1.get('throttled_usec',0)-s0.get('throttled_usec',0))/1e3:.0f} ms")
