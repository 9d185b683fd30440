import os, sys, subprocess
if len(sys.argv) > 1:
    sys.path.insert(0, os.getcwd())
    import numpy as np, deep_prove_amd as dpa
    P = 0xFFFFFFFF00000001
    nv = int(sys.argv[1])
    dev = dpa.Device(0)
    rng = np.random.default_rng(7)
    tabs = [rng.integers(0, P, size=1 << nv, dtype=np.uint64) for _ in range(2)]
    ms = [dpa.Mle.from_base(dev, t) for t in tabs]
    vp = dpa.VirtualPolynomial(nv); vp.add_mle_list(ms, (1, 0))
    proof, finals = dpa.prove_parallel(dev, vp, dpa.Transcript(b"test"))
    np.save(f"/tmp/proof_{os.environ.get('DP_NO_MULTI','0')}_{nv}.npy", proof)
else:
    import numpy as np
    for nv in (12, 13, 14):
        for nm in ("1", "0"):
            subprocess.run([sys.executable, __file__, str(nv)], env=dict(os.environ, DP_NO_MULTI=nm), check=True)
        a, b = np.load(f"/tmp/proof_1_{nv}.npy"), np.load(f"/tmp/proof_0_{nv}.npy")
        d = np.nonzero(a != b)[0]
        # stream: [npoint, point(2*nv words), nrounds, then per round: len(=3), 6 words]
        first = int(d[0]) if d.size else -1
        rnd = (first - (2 + 2 * nv)) // 7 if first >= 0 else -1
        print(f"nv={nv}: first differing word {first} -> round {rnd}; words differing {d.size} of {a.size}")
