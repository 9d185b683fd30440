"""Sweep of the cohort-regime knobs on one GPU, torch-free, every configuration in a fresh process (the knobs are read at
library / context creation) and every measured batch checked (proof 0 == the sequential proof, sampled proofs verify).
usage: python tools/knob_sweep.py [workload] [out.jsonl] [budget_s]          (driver; KNOB_ONLY=name,name restricts it)
       python tools/knob_sweep.py --one workload conc                        (one configuration, env = knobs)
Results: one JSON line per configuration, appended as they finish."""
import json, os, subprocess, sys, time

CONFIGS = [  # (name, proofs in flight, env) — in order of importance: the sweep stops when its time budget is spent. The sweeps of rounds 2 / 3 (results:
    # profiles/r02_*knob_sweep*.jsonl, r03_launch_count_sweeps.txt) covered knobs that round 4 removed with their code; what is left to sweep:
    ("base_448", 448, {}),
    ("base_256", 256, {}),
    ("cohort0_256", 256, {"DP_COHORT": "0"}),   # every proof on its own stream (round 1's scheme) with the fused tails
    ("cohort12_256", 256, {"DP_COHORT": "12"}),
    ("cohort32_448", 448, {"DP_COHORT": "32"}),
    ("hostsponge_448", 448, {"DP_HOST_SPONGE": "1"}),   # fused kernels, sponge on the host, requests served by every waiting thread (csrc/sponge_host.h)
    ("hostfs_256", 256, {"DP_DEVICE_FS": "0"}),         # sponge on the host: persistent sumchecks poll a mailbox per round, no fused tails (round 1's protocol kernels)
    ("nofused_448", 448, {"DP_DEVICE_LOGUP": "0", "DP_FUSED_OFF": "classic,dense,eqsum,commit,deleg"}),
    ("devlogup_tail_448", 448, {"DP_DEVICE_LOGUP": "1"}),   # the layer loop of every logup proof in one launch (2 = the whole logup proof)
    ("threads7_448", 448, {"DP_HOST_THREADS": "7"}),
]


def one(wl, conc):
    os.environ.setdefault("GPU_MAX_HW_QUEUES", "24")
    sys.path.insert(0, os.getcwd())
    import numpy as np
    import deep_prove_amd as dpa
    dev = dpa.Device(0)
    mb = getattr(dpa.models, wl)()
    ctx = dpa.Context.generate(dev, mb.blob())
    pr = dpa.Prover(ctx)
    waves = int(os.environ.get("KNOB_WAVES", "2"))  # proofs per measured batch = waves x in flight
    xs = np.stack([mb.input(3000 + i) for i in range(waves * conc)])
    single, out0 = pr.prove(xs[0])
    t0 = time.perf_counter(); pr.prove(xs[0]); lat = time.perf_counter() - t0
    pr.prove_batch(xs[:conc], conc)
    best = 0.0
    for _ in range(2):
        t0 = time.perf_counter(); proofs, outs, _ = pr.prove_batch(xs, conc); dt = time.perf_counter() - t0
        best = max(best, len(xs) / dt)
    ok = proofs[0].size == single.size and bool((proofs[0] == single).all())
    rec = {"proofs_per_s": round(best, 2), "waves": waves, "in_flight": pr.in_flight(), "latency_ms": round(1000 * lat, 2), "batch0_equals_single": ok}
    if not ok:  # where the experimental path leaves the validated one: first differing word of the canonical stream
        n = min(proofs[0].size, single.size)
        d = np.nonzero(proofs[0][:n] != single[:n])[0]
        rec.update({"size_batch": int(proofs[0].size), "size_single": int(single.size), "first_diff": int(d[0]) if d.size else n, "ndiff": int(d.size)})
    vb = ctx.verifier_blob()
    nver = 0
    try:
        for j in (1, len(xs) - 1):
            dpa.verify(vb, proofs[j], xs[j], outs[j]); nver += 1
    except Exception as e:  # noqa: BLE001
        rec["verify_error"] = f"{type(e).__name__}: {e}"[:200]
    rec["verified"] = nver
    print(json.dumps(rec), flush=True)


def main():
    if len(sys.argv) > 1 and sys.argv[1] == "--one":
        return one(sys.argv[2], int(sys.argv[3]))
    wl = sys.argv[1] if len(sys.argv) > 1 else "dense_4m"
    out = sys.argv[2] if len(sys.argv) > 2 else "gpurun_out/knob_sweep.jsonl"
    budget = float(sys.argv[3]) if len(sys.argv) > 3 else 300.0
    os.makedirs(os.path.dirname(out) or ".", exist_ok=True)
    t0 = time.time()
    only = set(filter(None, os.environ.get("KNOB_ONLY", "").split(",")))  # KNOB_ONLY=name,name: just these configurations
    for name, conc, env in CONFIGS:
        if only and name not in only:
            continue
        if time.time() - t0 > budget:
            break
        e = dict(os.environ); e.update(env)
        try:
            r = subprocess.run([sys.executable, os.path.abspath(__file__), "--one", wl, str(conc)], env=e, capture_output=True, text=True, timeout=int(os.environ.get('KNOB_TIMEOUT', '90')))
            line = r.stdout.strip().splitlines()[-1] if r.stdout.strip() else ""
            rec = json.loads(line) if line.startswith("{") else {"error": (r.stderr or r.stdout)[-400:]}
        except subprocess.TimeoutExpired:
            rec = {"error": "timeout"}
        rec.update({"config": name, "workload": wl, "asked": conc, "env": env, "t": round(time.time() - t0, 1)})
        with open(out, "a") as f:
            f.write(json.dumps(rec) + "\n")
        print(json.dumps(rec), flush=True)


if __name__ == "__main__":
    main()
