"""first contact of the resident executor with hardware: a small MLP batch under DP_RX=1 against the sequential proofs, then
Dense-4M throughput (DP_RX=1 vs cohorts). Run under `timeout -s KILL`: a protocol bug shows as a hang."""
import os, sys, time
sys.path.insert(0, os.getcwd())
import numpy as np
import deep_prove_amd as dpa
stage = sys.argv[1] if len(sys.argv) > 1 else "small"
dev = dpa.Device(0)
if stage == "small":
    mb = dpa.models.mlp(2, 64, config=43)
    ctx = dpa.Context.generate(dev, mb.blob())
    pr = dpa.Prover(ctx)
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 8
    xs = np.stack([mb.input(5000 + i) for i in range(n)])
    os.environ["DP_RX"] = "0"
    seq = [pr.prove(x) for x in xs]
    print("sequential proofs done", flush=True)
    os.environ["DP_RX"] = "1"
    t0 = time.perf_counter()
    proofs, outs, _ = pr.prove_batch(xs, n)
    print(f"rx batch of {n}: {1000 * (time.perf_counter() - t0):.1f} ms", flush=True)
    bad = [i for i in range(n) if proofs[i].size != seq[i][0].size or not (proofs[i] == seq[i][0]).all()]
    print("MISMATCH" if bad else "RX_PARITY_OK", bad, flush=True)
else:
    mb = dpa.models.dense_4m()
    ctx = dpa.Context.generate(dev, mb.blob())
    pr = dpa.Prover(ctx)
    pr.prove(mb.input(1000))
    conc = int(sys.argv[2]) if len(sys.argv) > 2 else 256
    waves = int(sys.argv[3]) if len(sys.argv) > 3 else 3
    xs = np.stack([mb.input(1000 + i) for i in range(conc * waves)])
    for mode in (sys.argv[4].split(",") if len(sys.argv) > 4 else ["1", "0"]):
        os.environ["DP_RX"] = mode
        pr.prove_batch(xs[:conc], conc)  # warm: workers, arenas
        t0 = time.perf_counter()
        proofs, outs, _ = pr.prove_batch(xs, conc)
        dt = time.perf_counter() - t0
        v, _ = dpa.verify_batch(ctx.verifier_blob(), proofs[:32], xs[:32], outs[:32], dev=dev)
        print(f"DP_RX={mode}: {len(xs) / dt:.1f} proofs/s ({conc} in flight, {len(xs)} proofs, {1000 * dt:.0f} ms), rejected of 32: {int(v.sum())}", flush=True)
