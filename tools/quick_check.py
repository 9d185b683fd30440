"""One short GPU pass without torch (a fresh box spends 1-2 min importing it): golden sha256 of a single Dense-4M / CNN-264k
proof, throughput at the bench's default number of proofs in flight, batch proofs == sequential proofs, host verifier.
usage: python tools/quick_check.py [out.json] [budget_s]; results are flushed stage by stage."""
import hashlib, json, os, sys, time
T0 = time.time()
os.environ.setdefault("GPU_MAX_HW_QUEUES", "24")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import deep_prove_amd as dpa
out_path = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "gpurun_out", "quick.json")
budget = float(sys.argv[2]) if len(sys.argv) > 2 else 110.0
os.makedirs(os.path.dirname(out_path), exist_ok=True)
res = {"stages": []}


def emit(**kw):
    kw["t"] = round(time.time() - T0, 2)
    res["stages"].append(kw)
    json.dump(res, open(out_path, "w"), indent=1)
    print(json.dumps(kw), flush=True)


dev = dpa.Device(0)
emit(stage="device", name=dev.name)
for wl, gold_name, specs in (("dense_4m", "dense4m_proof.json", [(192, 8), (192, 16), (256, 8), (96, 8)]),
                             ("cnn_264k", "cnn264k_proof.json", [(192, 8), (96, 8)])):
    if time.time() - T0 > budget:
        break
    gold = json.load(open(os.path.join(ROOT, "tests", "golden", gold_name)))
    mb = getattr(dpa.models, wl)()
    ctx = dpa.Context.generate(dev, mb.blob())
    pr = dpa.Prover(ctx)
    vb = ctx.verifier_blob()
    x = mb.input(gold["input_index"])
    t0 = time.perf_counter(); proof, out = pr.prove(x); first = time.perf_counter() - t0
    t0 = time.perf_counter(); proof, out = pr.prove(x); lat = time.perf_counter() - t0
    emit(stage="single", workload=wl, sha_ok=hashlib.sha256(proof.tobytes()).hexdigest() == gold["sha256"], first_ms=round(1000 * first, 1), latency_ms=round(1000 * lat, 1))
    for conc, co in specs:
        if time.time() - T0 > budget:
            break
        os.environ["DP_COHORT"] = str(co)
        xs = np.stack([x] + [mb.input(3000 + i) for i in range(2 * conc - 1)])
        try:
            pr.prove_batch(xs[:conc], conc)  # creates the workers
            t0 = time.perf_counter(); proofs, outs, _ = pr.prove_batch(xs, conc); dt = time.perf_counter() - t0
            same = proofs[0].size == proof.size and bool((proofs[0] == proof).all())
            for j in (1, len(xs) // 2, len(xs) - 1):
                dpa.verify(vb, proofs[j], xs[j], outs[j])
            emit(stage="batch", workload=wl, asked=conc, cohort=co, in_flight=pr.in_flight(), proofs=len(xs), proofs_per_s=round(len(xs) / dt, 2), batch0_equals_single=same, verified=3)
            del proofs
        except Exception as e:  # noqa: BLE001
            emit(stage="batch", workload=wl, asked=conc, cohort=co, error=f"{type(e).__name__}: {e}")
            break
    ctx.free()
emit(stage="done")
