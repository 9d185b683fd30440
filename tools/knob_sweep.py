"""Sweep of the cohort-regime knobs on one GPU, torch-free, every configuration in a fresh process (the knobs are read at
library / context creation) and every measured batch checked (proof 0 == the sequential proof, sampled proofs verify).
usage: python tools/knob_sweep.py [workload] [out.jsonl] [budget_s]          (driver; KNOB_ONLY=name,name restricts it)
       python tools/knob_sweep.py --one workload conc                        (one configuration, env = knobs)
Results: one JSON line per configuration, appended as they finish."""
import json, os, subprocess, sys, time

CONFIGS = [  # (name, proofs in flight, env) — in order of importance: the sweep stops when its time budget is spent
    ("base_192", 192, {}),
    ("base_256", 256, {}),
    # more proofs in flight (the twiddle tables are shared since round 2: 600 MB per proof in flight)
    # the fused commit tail taking over earlier (default: folded oracle <= 4096)
    ("ctail16k_256", 256, {"DP_COMMIT_TAIL_MAX_N": "16384"}),
    ("ctail64k_256", 256, {"DP_COMMIT_TAIL_MAX_N": "65536"}),
    ("ctail8k_256", 256, {"DP_COMMIT_TAIL_MAX_N": "8192"}),
    ("ctail32k_256", 256, {"DP_COMMIT_TAIL_MAX_N": "32768"}),
    ("c20_480_arena480", 480, {"DP_COHORT": "20", "DP_WORKER_ARENA_BYTES": str(480 << 20)}),
    ("hostsponge_256", 256, {"DP_HOST_SPONGE": "1"}),   # fused kernels, sponge on the host, requests served by every waiting thread (csrc/sponge_host.h)
    ("hostsponge_s8_256", 256, {"DP_HOST_SPONGE": "1", "DP_SPONGE_THREADS": "8"}),
    ("hostsponge_s4_256", 256, {"DP_HOST_SPONGE": "1", "DP_SPONGE_THREADS": "4"}),
    ("hostfs_256", 256, {"DP_DEVICE_FS": "0"}),   # sponge on the host: persistent sumchecks poll a mailbox per round, no fused tails (round 1's protocol kernels)
    ("cohort0_256", 256, {"DP_COHORT": "0"}),   # every proof on its own stream (round 1's scheme) with round 2's fused tails
    ("cohort4_256", 256, {"DP_COHORT": "4"}),
    ("cohort6_256", 256, {"DP_COHORT": "6"}),
    ("cohort6_256_q32", 256, {"DP_COHORT": "6", "GPU_MAX_HW_QUEUES": "32"}),
    ("c16_384", 384, {"DP_COHORT": "16"}),
    ("c22_512", 512, {"DP_COHORT": "22"}),
    ("c14_320", 320, {"DP_COHORT": "14"}),
    ("c12_384", 384, {"DP_COHORT": "12", "GPU_MAX_HW_QUEUES": "32"}),
    # round 2: one-workgroup kernels shared (256 threads, no CU reservation, raised wave priority) vs whole-CU workgroups
    ("excl_192", 192, {"DP_SHARED_TAILS": "0"}),
    ("cohort8_256", 256, {"DP_COHORT": "8"}),
    ("noasync_256", 256, {"DP_ASYNC_UPLOAD": "0"}),
    ("fuse4_256", 256, {"DP_MERKLE_FUSE": "4"}),
    ("tailmax2048_256", 256, {"DP_TAIL_MAX": "2048"}),
    ("fuse4_tailmax2048_256", 256, {"DP_MERKLE_FUSE": "4", "DP_TAIL_MAX": "2048"}),
    ("fuse4_tailmax1024_256", 256, {"DP_MERKLE_FUSE": "4", "DP_TAIL_MAX": "1024"}),
    ("fuse8_tailmax2048_256", 256, {"DP_MERKLE_FUSE": "8", "DP_TAIL_MAX": "2048"}),
    ("cohort16_fuse4_tailmax2048_256", 256, {"DP_COHORT": "16", "DP_MERKLE_FUSE": "4", "DP_TAIL_MAX": "2048"}),
    ("cohort12_256", 256, {"DP_COHORT": "12"}),
    ("cohort16_t16_256", 256, {"DP_COHORT": "16", "DP_HOST_THREADS": "16"}),
    ("cohort24_256", 256, {"DP_COHORT": "24"}),
    ("shared512_256", 256, {"DP_SHARED_THREADS": "512"}),
    ("shared128_256", 256, {"DP_SHARED_THREADS": "128"}),
    ("cohort32_256", 256, {"DP_COHORT": "32"}),
    ("cohort64_256", 256, {"DP_COHORT": "64"}),
    ("cohort32_256_t8", 256, {"DP_COHORT": "32", "DP_HOST_THREADS": "8"}),
    ("cohort16_256_q16", 256, {"DP_COHORT": "16", "GPU_MAX_HW_QUEUES": "16"}),
    ("cohort128_256", 256, {"DP_COHORT": "128"}),
    ("skiphash_256", 256, {"DP_DEBUG_SKIP_HASH": "1"}),  # timing experiment: wide Merkle layers without the Poseidon2 work (proofs do not verify)
    ("skiphash_192", 192, {"DP_DEBUG_SKIP_HASH": "1"}),
    ("threads7_256", 256, {"DP_HOST_THREADS": "7"}),
    # wide kernels capped to a few workgroups per CU (grid-stride loops): do free wave slots shorten the queueing of the small kernels?
    ("maxgrid32_256", 256, {"DP_MAX_GRID": "32"}),
    ("maxgrid64_256", 256, {"DP_MAX_GRID": "64"}),
    ("maxgrid128_256", 256, {"DP_MAX_GRID": "128"}),
    ("maxgrid256_256", 256, {"DP_MAX_GRID": "256"}),
    ("maxgrid64_lp256_256", 256, {"DP_MAX_GRID": "64", "DP_MERKLE_LP_MAX": "256"}),
    ("lp256_256", 256, {"DP_MERKLE_LP_MAX": "256"}),
    ("lp1024_256", 256, {"DP_MERKLE_LP_MAX": "1024"}),
    ("shared128_192", 192, {"DP_SHARED_THREADS": "128"}),
    ("shared512_192", 192, {"DP_SHARED_THREADS": "512"}),
    ("shared1024_192", 192, {"DP_SHARED_THREADS": "1024"}),
    ("shared_cohort16_256", 256, {"DP_COHORT": "16"}),
    ("shared_cohort4_192", 192, {"DP_COHORT": "4"}),
    ("shared_nofused_192", 192, {"DP_DEVICE_LOGUP": "0", "DP_DEVICE_CLASSIC": "0", "DP_DEVICE_DENSE": "0", "DP_DEVICE_EQSUM": "0", "DP_DEVICE_COMMIT": "0"}),
    # k_logup_tail: written after round 1's GPU budget ran out, validated on the CPU SIMT emulator only (tests/test_kernel_emul.py)
    ("devlogup_tail_192", 192, {"DP_DEVICE_LOGUP": "1"}),   # the layer loop of every logup proof in one launch
    ("devlogup_full_192", 192, {"DP_DEVICE_LOGUP": "2"}),   # the whole logup proof (trees, outputs, layers, column claims) in one launch
    ("devlogup_full_256", 256, {"DP_DEVICE_LOGUP": "2"}),
    ("devclassic_192", 192, {"DP_DEVICE_CLASSIC": "1"}),   # k_classic_tail: the last rounds of the batch-opening sumcheck in one launch
    ("devcommit_192", 192, {"DP_DEVICE_COMMIT": "1"}),     # k_commit_tail: the last rounds of the Basefold commit phase (fold, merges, Merkle, roots) in one launch
    ("deveqsum_192", 192, {"DP_DEVICE_EQSUM": "1"}),       # k_eqsum_tail: the eq tables + accumulation sumcheck of Requant / ReLU in one launch
    ("devdense_192", 192, {"DP_DEVICE_DENSE": "1"}),       # k_dense_tail: bias evaluation + fix_high + sumcheck of a Dense layer in one launch
    ("devall_192", 192, {"DP_DEVICE_LOGUP": "2", "DP_DEVICE_CLASSIC": "1", "DP_DEVICE_DENSE": "1", "DP_DEVICE_EQSUM": "1", "DP_DEVICE_COMMIT": "1"}),
    ("devall_256", 256, {"DP_DEVICE_LOGUP": "2", "DP_DEVICE_CLASSIC": "1", "DP_DEVICE_DENSE": "1", "DP_DEVICE_EQSUM": "1", "DP_DEVICE_COMMIT": "1"}),
    ("async_upload_192", 192, {"DP_ASYNC_UPLOAD": "1"}),  # uploads take ring slots of the staging buffer instead of a copy + publish + wait each (~29 per proof)
    ("devall_async_tailmax2048_192", 192, {"DP_DEVICE_LOGUP": "2", "DP_DEVICE_CLASSIC": "1", "DP_DEVICE_DENSE": "1", "DP_DEVICE_EQSUM": "1", "DP_DEVICE_COMMIT": "1", "DP_ASYNC_UPLOAD": "1", "DP_TAIL_MAX": "2048", "DP_MERKLE_FUSE": "4"}),
    ("devall_async_tailmax2048_256", 256, {"DP_DEVICE_LOGUP": "2", "DP_DEVICE_CLASSIC": "1", "DP_DEVICE_DENSE": "1", "DP_DEVICE_EQSUM": "1", "DP_DEVICE_COMMIT": "1", "DP_ASYNC_UPLOAD": "1", "DP_TAIL_MAX": "2048", "DP_MERKLE_FUSE": "4"}),
    # fewer launches per Merkle tree: one workgroup finishes every tree from 2048 / 4096 digests on (4 - 5 launches less per tree,
    # ~35 trees per Dense-4M proof); validated code path (k_merkle_tail), the default 256 is the single-proof latency optimum
    ("merklefuse4_192", 192, {"DP_MERKLE_FUSE": "4"}),   # k_merkle_layers: four Merkle layers per launch (emulator-validated, new on hardware)
    ("merklefuse4_tailmax2048_192", 192, {"DP_MERKLE_FUSE": "4", "DP_TAIL_MAX": "2048"}),
    ("tailmax2048_192", 192, {"DP_TAIL_MAX": "2048"}),
    ("tailmax4096_192", 192, {"DP_TAIL_MAX": "4096"}),
    ("devall_tailmax2048_192", 192, {"DP_DEVICE_LOGUP": "2", "DP_DEVICE_CLASSIC": "1", "DP_DEVICE_DENSE": "1", "DP_DEVICE_EQSUM": "1", "DP_DEVICE_COMMIT": "1", "DP_TAIL_MAX": "2048"}),
    ("devall_tailmax2048_256", 256, {"DP_DEVICE_LOGUP": "2", "DP_DEVICE_CLASSIC": "1", "DP_DEVICE_DENSE": "1", "DP_DEVICE_EQSUM": "1", "DP_DEVICE_COMMIT": "1", "DP_TAIL_MAX": "2048"}),
    ("threads7_192", 192, {"DP_HOST_THREADS": "7"}),
    ("threads4_192", 192, {"DP_HOST_THREADS": "4"}),
    ("tail256_192", 192, {"DP_TAIL_MANY_THREADS": "256", "DP_TAIL_MANY_EXCL": "0"}),
    ("tail512_192", 192, {"DP_TAIL_MANY_THREADS": "512", "DP_TAIL_MANY_EXCL": "0"}),
    ("tailnoexcl_192", 192, {"DP_TAIL_MANY_EXCL": "0"}),
    ("cohort_noexcl_192", 192, {"DP_COHORT_EXCL": "0"}),
    ("cohort_noexcl_tail256_192", 192, {"DP_COHORT_EXCL": "0", "DP_TAIL_MANY_THREADS": "256", "DP_TAIL_MANY_EXCL": "0"}),
    ("noexcl_192", 192, {"DP_NO_EXCLUSIVE_CU": "1"}),
    ("cohort4_192", 192, {"DP_COHORT": "4"}),
    ("cohort12_192", 192, {"DP_COHORT": "12"}),
    ("tail256_256", 256, {"DP_TAIL_MANY_THREADS": "256", "DP_TAIL_MANY_EXCL": "0"}),
    ("xcd_192", 192, {"DP_COHORT_XCD": "1"}),
    ("xcd_256", 256, {"DP_COHORT_XCD": "1"}),
    ("persist256_noexcl_192", 192, {"DP_COHORT_EXCL": "0", "DP_COHORT_PERSIST_THREADS": "256"}),
    ("persist512_192", 192, {"DP_COHORT_PERSIST_THREADS": "512"}),
    ("hostfs_192", 192, {"DP_DEVICE_FS": "0"}),
    ("lpmax256_192", 192, {"DP_MERKLE_LP_MAX": "256"}),
    ("tailmax1024_192", 192, {"DP_TAIL_MAX": "1024"}),
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
