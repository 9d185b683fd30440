"""GPU parity of the sharded sumcheck (BASELINE config 5): W device contexts each own a contiguous slice of every
table and run dp_sc_session_* on it; the shares are combined by deep_prove_amd.sharded.prove_sharded. The proof must be
bit-identical to the unsharded device prover and to the oracle."""
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
