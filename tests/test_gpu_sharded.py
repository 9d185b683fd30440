"""GPU parity of the sharded sumcheck (BASELINE config 5): W device contexts each own a contiguous slice of every
table and run dp_sc_session_* on it; the shares are combined by deep_prove_amd.sharded.prove_sharded. The proof must be
bit-identical to the unsharded device prover and to the oracle."""
import os
import numpy as np
import pytest

pytestmark = pytest.mark.gpu
P = 0xFFFFFFFF00000001


def run_sharded(dpa, devs, tabs, terms, nv):
    world = len(devs)
    chunk = (1 << nv) // world
    k = world.bit_length() - 1
    mles, shards = [], []
    for g, d in enumerate(devs):
        ms = [dpa.Mle.from_base(d, t[g * chunk:(g + 1) * chunk]) for t in tabs]
        mles += ms
        shards.append(dpa.sharded.HipShard(d, nv - k, ms, terms))
    small_mles = []

    def make_small(table_words):
        ms = [dpa.Mle.from_ext(devs[0], w) for w in table_words]
        small_mles.extend(ms)
        return dpa.sharded.HipShard(devs[0], k, ms, terms)

    ex = dpa.sharded.LocalExchange(world)
    proof, finals = dpa.sharded.prove_sharded(shards, ex, nv, terms, dpa.Transcript(b"test"), make_small)
    for s in shards:
        s.close()
    for m in mles + small_mles:
        m.free()
    return proof, finals


@pytest.mark.parametrize("world,nv", [(1, 10), (2, 10), (4, 12), (2, 16)])
def test_sharded_sumcheck_matches_unsharded_device_and_oracle(oracle, world, nv):
    import deep_prove_amd as dpa
    rng = np.random.default_rng(100 + nv + world)
    tabs = [rng.integers(0, P, size=1 << nv, dtype=np.uint64) for _ in range(3)]
    terms = [((1, 0), [0, 1, 2]), ((5, 7), [1, 2])]
    devs = [dpa.Device(0) for _ in range(world)]
    try:
        proof, finals = run_sharded(dpa, devs, tabs, terms, nv)
        # unsharded device prover
        ms = [dpa.Mle.from_base(devs[0], t) for t in tabs]
        vp = dpa.VirtualPolynomial(nv)
        vp.add_mle_list(ms, (1, 0))
        vp.add_mle_list(ms[1:], (5, 7))
        dproof, dfinals = dpa.prove_parallel(devs[0], vp, dpa.Transcript(b"test"))
        for m in ms:
            m.free()
    finally:
        for d in devs:
            d.close()
    oproof, ofinals = oracle.sumcheck_prove(nv, tabs, [False] * 3, terms, oracle.transcript(b"test"))
    assert proof.size == oproof.size and (proof == oproof).all()
    assert (finals == ofinals).all()
    assert (dproof == oproof).all() and (dfinals == ofinals).all()


def test_sharded_2pow22_roundtrip_properties():
    """larger than the oracle comfortably handles: W = 4 slices of a 2^22 product of 3 base tables; the sharded proof equals
    the unsharded device proof (size-independent property: same transcript, same message stream)"""
    import deep_prove_amd as dpa
    nv, world = 22, 4
    tabs = [dpa.models.splitmix64(0xD33B0000 ^ (5 << 32) ^ j, 1 << nv) % np.uint64(P) for j in range(3)]
    terms = [((1, 0), [0, 1, 2])]
    devs = [dpa.Device(0) for _ in range(world)]
    try:
        proof, finals = run_sharded(dpa, devs, tabs, terms, nv)
        ms = [dpa.Mle.from_base(devs[0], t) for t in tabs]
        vp = dpa.VirtualPolynomial(nv)
        vp.add_mle_list(ms, (1, 0))
        dproof, dfinals = dpa.prove_parallel(devs[0], vp, dpa.Transcript(b"test"))
    finally:
        for d in devs:
            d.close()
    assert proof.size == dproof.size and (proof == dproof).all() and (finals == dfinals).all()


@pytest.mark.parametrize("world,nv", [(1, 9), (2, 10), (4, 12), (8, 13), (4, 18)])
def test_in_library_sharded_loop_on_several_contexts(oracle, world, nv):
    """dp_sumcheck_prove_sharded_local: the C++ round loop of csrc/sharded.h with `world` device contexts on one GPU, one thread
    per rank — local round sums on the device, exchange, mod-p sum and sponge in the library, stage 2 on W-entry tables —
    bit-identical to the oracle's unsharded proof (and every rank's proof equals rank 0's: checked inside the call)"""
    import deep_prove_amd as dpa
    rng = np.random.default_rng(300 + nv + world)
    tabs = [rng.integers(0, P, size=(2 if e else 1) << nv, dtype=np.uint64) for e in (False, True, False)]
    exts = [False, True, False]
    terms = [((1, 0), [0, 1, 2]), ((5, 7), [1, 2]), ((9, 0), [0])]
    chunk = (1 << nv) // world
    devs = [dpa.Device(0) for _ in range(world)]
    try:
        rows = [[(dpa.Mle.from_ext(d, t[2 * g * chunk:2 * (g + 1) * chunk]) if e else dpa.Mle.from_base(d, t[g * chunk:(g + 1) * chunk])) for t, e in zip(tabs, exts)]
                for g, d in enumerate(devs)]
        ts = [dpa.Transcript(b"test") for _ in range(world)]
        proof, finals = dpa.sharded.prove_sharded_local(devs, nv, rows, terms, ts)
        after = [t.read_challenge() for t in ts]
        for row in rows:
            for m in row:
                m.free()
    finally:
        for d in devs:
            d.close()
    ot = oracle.transcript(b"test")
    oproof, ofinals = oracle.sumcheck_prove(nv, tabs, exts, terms, ot)
    assert proof.size == oproof.size and (proof == oproof).all() and (finals == ofinals).all()
    assert all(a == after[0] for a in after) and after[0] == ot.read_challenge()  # every rank's transcript ends where the oracle's does


def test_in_library_sharded_loop_over_rccl_world_of_one(dev, oracle):
    """dp_dist_unique_id / dp_dist_init / dp_sumcheck_prove_sharded with a REAL RCCL communicator (librccl through dlopen,
    ncclCommInitRank, ncclAllGather on device buffers) — a world of one is what a 1-GPU box can run; it exercises every call of
    the RCCL path except the wire"""
    import ctypes as C
    import deep_prove_amd as dpa
    from deep_prove_amd import _lib
    lib = _lib.load()
    ident = (C.c_uint8 * 128)()
    _lib.check(lib.dp_dist_unique_id(ident))
    h = _lib.vp()
    _lib.check(lib.dp_dist_init(dev.h, ident, 0, 1, C.byref(h)))

    class G:
        pass
    g = G(); g.h = h
    nv = 14
    rng = np.random.default_rng(77)
    tabs = [rng.integers(0, P, size=1 << nv, dtype=np.uint64) for _ in range(3)]
    terms = [((1, 0), [0, 1, 2]), ((3, 4), [2, 0])]
    ms = [dpa.Mle.from_base(dev, t) for t in tabs]
    try:
        proof, finals = dpa.sharded.prove_sharded_in_library(dev, g, nv, ms, terms, dpa.Transcript(b"test"))
        solo, sfinals = dpa.sharded.prove_sharded_in_library(dev, None, nv, ms, terms, dpa.Transcript(b"test"))
    finally:
        for m in ms:
            m.free()
        _lib.check(lib.dp_dist_free(h))
    oproof, ofinals = oracle.sumcheck_prove(nv, tabs, [False] * 3, terms, oracle.transcript(b"test"))
    assert (proof == oproof).all() and (finals == ofinals).all() and (solo == oproof).all() and (sfinals == ofinals).all()


def _golden_sc(nv):
    import json
    import os
    g = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "sumcheck24.json")))
    return g["cases"][str(nv)], g["k"]


def _check_against_golden(dpa, gold, nv, proof, finals, transcript):
    import hashlib
    assert proof.size == gold["proof_words"]
    assert hashlib.sha256(proof.tobytes()).hexdigest() == gold["sha256"], f"2^{nv} sumcheck: proof stream differs from the oracle's"
    assert [int(v) for v in finals] == gold["finals"]
    assert list(transcript.read_challenge()) == gold["next_challenge"]  # the sponge ends where the oracle's does
    # and the host verifier accepts it: claimed sum = p_0(0) + p_0(1) of the first round message, sub-claim = product of the finals
    off = 1 + 2 * nv + 1 + 1
    e0, e1 = (int(proof[off]), int(proof[off + 1])), (int(proof[off + 2]), int(proof[off + 3]))
    P_ = 0xFFFFFFFF00000001
    claimed = ((e0[0] + e1[0]) % P_, (e0[1] + e1[1]) % P_)
    point, expected = dpa.verify_sumcheck(claimed, proof, nv, 3, dpa.Transcript(b"test"))

    def emul(a, b):
        return ((a[0] * b[0] + 7 * a[1] * b[1]) % P_, (a[0] * b[1] + a[1] * b[0]) % P_)
    prod = (1, 0)
    for i in range(3):
        prod = emul(prod, (int(finals[2 * i]), int(finals[2 * i + 1])))
    assert prod == expected, "final evaluations do not multiply to the verifier's sub-claim"


@pytest.mark.parametrize("nv", [22, 24])
def test_config5_prove_parallel_at_size_equals_oracle_golden(dev, nv):
    """BASELINE config 5 at its stated size (and 2^22): the single-GPU prove_parallel bench.py times — streaming rounds with the
    t = 1 skip, the hand-over to the LDS session, the k_reduce_publish chain — against the oracle's committed sha256
    (tests/golden/sumcheck24.json, tests/golden/make_sumcheck24_hash.py; shape of sumcheck/benches/devirgo_sumcheck.rs:42-101)"""
    import deep_prove_amd as dpa
    gold, k = _golden_sc(nv)
    tabs = [dpa.Mle.from_base(dev, dpa.models.splitmix64(0xD33B0000 ^ (5 << 32) ^ j, 1 << nv) % np.uint64(P)) for j in range(k)]
    try:
        vp = dpa.VirtualPolynomial(nv)
        vp.add_mle_list(tabs, (1, 0))
        t = dpa.Transcript(b"test")
        proof, finals = dpa.prove_parallel(dev, vp, t)
    finally:
        for m in tabs:
            m.free()
    _check_against_golden(dpa, gold, nv, proof, finals, t)


def test_config5_round_by_round_form_equals_oracle_golden():
    """DP_SC_GRID2=0: the round-by-round streaming form (k_sc_terms, then one k_sc_fused launch per round) that the two-round grid replaced as the default for large
    base-table products stays the fallback — same golden sha256 at 2^22, in its own process (the knob is read once)"""
    import subprocess
    import sys
    gold, _ = _golden_sc(22)
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for knob in ("0", "1"):
        r = subprocess.run([sys.executable, os.path.join(root, "tools", "sumcheck24_only.py"), "2", "22"], capture_output=True, text=True, timeout=300, cwd=root,
                           env=dict(os.environ, DP_SC_GRID2=knob, SC24_PROFILE="1"))
        assert r.returncode == 0, r.stderr[-2000:]
        shas = [ln.split()[-1] for ln in r.stdout.splitlines() if "proof sha256" in ln]
        assert len(shas) == 2 and all(h == gold["sha256"][:16] for h in shas), (knob, r.stdout)
        assert ("k_sc_terms2" in r.stdout) == (knob == "1"), (knob, r.stdout)  # the form that ran is the one asked for


def test_fused_round_ticket_under_reuse_of_its_partial_buffer(dev):
    """k_sc_fused's in-kernel "last workgroup" reduction (kernels.inc: block sums written through, a relaxed device-scope ticket, no fence — the default since round 4;
    DP_FUSED_TICKET=0 is the fallback with a separate k_reduce_publish launch) relies on write-through / L2-bypass behaviour of sc0 sc1 accesses that the HIP memory model
    does not spell out. Stress: the same 2^22 sumcheck 40 times back to back — every repetition reuses the SAME partial-sum buffer and ticket, the last workgroup lands
    on a different XCD from launch to launch, stale lines of the previous repetition sit in eight L2s — and every proof must equal the oracle's golden (advisor, round 4)."""
    import hashlib
    import deep_prove_amd as dpa
    nv = 22
    gold, k = _golden_sc(nv)
    tabs = [dpa.Mle.from_base(dev, dpa.models.splitmix64(0xD33B0000 ^ (5 << 32) ^ j, 1 << nv) % np.uint64(P)) for j in range(k)]
    try:
        vp = dpa.VirtualPolynomial(nv)
        vp.add_mle_list(tabs, (1, 0))
        for rep in range(40):
            proof, finals = dpa.prove_parallel(dev, vp, dpa.Transcript(b"test"))
            assert hashlib.sha256(proof.tobytes()).hexdigest() == gold["sha256"], f"repetition {rep}: the proof differs from the oracle's"
    finally:
        for m in tabs:
            m.free()


@pytest.mark.parametrize("nv,world", [(22, 4), (24, 8)])
def test_config5_sharded_in_library_at_size_equals_oracle_golden(nv, world):
    """the same sumcheck through dp_sumcheck_prove_sharded_local: `world` device contexts (one thread each) own contiguous slices,
    the round loop runs in the library — W = 8 at 2^24 is BASELINE config 5's partition on one GPU"""
    import deep_prove_amd as dpa
    gold, k = _golden_sc(nv)
    chunk = (1 << nv) // world
    full = [dpa.models.splitmix64(0xD33B0000 ^ (5 << 32) ^ j, 1 << nv) % np.uint64(P) for j in range(k)]
    devs = [dpa.Device(0) for _ in range(world)]
    try:
        rows = [[dpa.Mle.from_base(d, t[g * chunk:(g + 1) * chunk]) for t in full] for g, d in enumerate(devs)]
        ts = [dpa.Transcript(b"test") for _ in range(world)]
        proof, finals = dpa.sharded.prove_sharded_local(devs, nv, rows, [((1, 0), list(range(k)))], ts)
        for row in rows:
            for m in row:
                m.free()
    finally:
        for d in devs:
            d.close()
    _check_against_golden(dpa, gold, nv, proof, finals, ts[0])
