"""The submit / poll forms of the seam calls (dp_async, include/deep_prove_hip.h "asynchronous seam calls"): every call kind against its blocking form —
same proof words, same transcript afterwards — with many calls of identical shape in flight from ONE thread (they are proved in lock step, launches
merged: the engine's counters must say so), calls of different shapes interleaved, and against the oracle for one instance of each kind."""
import numpy as np
import pytest

from test_gpu_primitives import P, rand_base, rand_point

pytestmark = pytest.mark.gpu


def _wait_all(tickets):
    pending = list(tickets)
    while pending:  # a polling loop, as a single-threaded host would run it
        pending = [t for t in pending if t.poll() == 0]


def test_async_sumcheck_and_logup_match_the_blocking_calls_and_merge(dev, oracle):
    import deep_prove_amd as dpa
    rng = np.random.default_rng(4242)
    dpa.Basefold(dev, 1 << 16)  # dp_pcs_setup before the engine: its workers share the PCS tables
    eng = dpa.AsyncEngine(dev, max_in_flight=24, worker_arena_bytes=256 << 20)
    try:
        nv, n = 10, 1 << 10
        jobs = []
        for j in range(12):  # twelve sumchecks of one shape (two degree-2 products over three tables) + twelve lookups of one shape, interleaved
            tabs = [rand_base(rng, n) for _ in range(2)] + [rand_base(rng, 2 * n)]
            cols = [rng.integers(0, 1 << 20, size=n, dtype=np.uint64) for _ in range(2)]
            cc, csc = rand_point(rng, 1)[0], rand_point(rng, 1)[0]
            co = [rand_point(rng, 1)[0], rand_point(rng, 1)[0]]
            jobs.append((tabs, cols, cc, csc, co))
        def make_vp(tabs, co):
            m = [dpa.Mle.from_base(dev, tabs[0]), dpa.Mle.from_base(dev, tabs[1]), dpa.Mle.from_ext(dev, tabs[2])]
            vp = dpa.VirtualPolynomial(nv)
            vp.add_mle_list([m[0], m[1]], co[0]); vp.add_mle_list([m[0], m[2]], co[1])
            return vp
        # blocking reference
        ref = []
        for tabs, cols, cc, csc, co in jobs:
            t1, t2 = dpa.Transcript(b"async"), dpa.Transcript(b"async")
            pw, fin = dpa.prove_parallel(dev, make_vp(tabs, co), t1)
            lw = dpa.logup_batch_prove(dev, [dpa.Mle.from_base(dev, c) for c in cols], 2, cc, csc, t2)
            ref.append((pw, fin, t1.read_challenge(), lw, t2.read_challenge()))
        # the same calls, all in flight at once from this thread
        # (every table is uploaded BEFORE the first submit: the engine gathers calls of one shape for 100 us, and an upload between two submits is longer than that)
        prepared = [(dpa.Transcript(b"async"), dpa.Transcript(b"async"), make_vp(tabs, co), [dpa.Mle.from_base(dev, c) for c in cols], cc, csc) for tabs, cols, cc, csc, co in jobs]
        live, tickets = [], []
        for t1, t2, vp, mcols, cc, csc in prepared:
            k1 = eng.prove_parallel(vp, t1)
            k2 = eng.logup_batch_prove(mcols, 2, cc, csc, t2)
            live.append((t1, t2, k1, k2, vp)); tickets += [k1, k2]
        _wait_all(tickets)
        for (pw, fin, c1, lw, c2), (t1, t2, k1, k2, vp) in zip(ref, live):
            got = k1.words()
            assert got.size == pw.size and (got == pw).all()
            assert (k1.values(fin.size) == fin).all() and t1.read_challenge() == c1
            gl = k2.words()
            assert gl.size == lw.size and (gl == lw).all() and t2.read_challenge() == c2
            k1.free(); k2.free()
        st = eng.stats()
        # identical shapes queued together ran merged (how many of the 24 meet in one group is a matter of timing: the words above are the test, this only says the merged path ran)
        assert st["calls"] == 24 and st["merged_calls"] >= 2 and st["groups"] < 24, st
        # one instance against the oracle
        tabs, cols, cc, csc, co = jobs[0]
        ot = oracle.transcript(b"async")
        exp = oracle.logup_prove(cols, 2, cc, csc, ot, None)
        assert exp.size == ref[0][3].size and (exp == ref[0][3]).all()
    finally:
        eng.close()


def test_async_commit_and_batch_open_match_the_blocking_calls(dev, oracle):
    import deep_prove_amd as dpa
    rng = np.random.default_rng(777)
    maxsize = 1 << 14
    pcs = dpa.Basefold(dev, maxsize)
    eng = dpa.AsyncEngine(dev, max_in_flight=16, worker_arena_bytes=256 << 20)
    try:
        shape = [(12, False), (10, False), (10, True), (9, False)]
        sets = []
        for _ in range(4):  # four independent "proofs": commit four polynomials each, open them in one batch
            raws = [rand_base(rng, (2 if e else 1) << nv) for nv, e in shape]
            sets.append(raws)
        mk = lambda raws: [dpa.Mle.from_ext(dev, w) if e else dpa.Mle.from_base(dev, w) for w, (nv, e) in zip(raws, shape)]  # noqa: E731
        points = [rand_point(rng, nv) for nv, _ in shape]
        ref = []
        for raws in sets:
            mles = mk(raws)
            comms = [pcs.commit(m) for m in mles]
            evals = [m.evaluate(p) for m, p in zip(mles, points)]
            t = dpa.Transcript(b"open")
            ref.append(([c.root for c in comms], evals, pcs.batch_open(comms, points, evals, t), t.read_challenge()))
        # async: all 16 commits in flight, then the 4 openings in flight
        mless = [mk(raws) for raws in sets]
        cticks = [[eng.commit(m) for m in mles] for mles in mless]
        _wait_all([k for ks in cticks for k in ks])
        commss = [[k.commitment(dev, m) for k, m in zip(ks, mles)] for ks, mles in zip(cticks, mless)]
        for ks in cticks:
            for k in ks:
                k.free()
        for (roots, _, _, _), comms in zip(ref, commss):
            assert [c.root for c in comms] == roots
        ts = [dpa.Transcript(b"open") for _ in sets]
        oticks = [eng.batch_open(comms, points, r[1], t) for comms, r, t in zip(commss, ref, ts)]
        _wait_all(oticks)
        for k, t, (roots, evals, proof, chal) in zip(oticks, ts, ref):
            got = k.words()
            assert got.size == proof.size and (got == proof).all() and t.read_challenge() == chal
            k.free()
        # PCS::commit(&poly) with the polynomial on the host: upload + commit in one ticket
        hk = [eng.commit_host(w, e) for w, (nv, e) in zip(sets[1], shape)]
        _wait_all(hk)
        for k, want in zip(hk, ref[1][0]):
            tab = k.table(dev)
            c = k.commitment(dev, tab)
            assert c.root == want
            k.free()
        # against the oracle and the host verifier
        roots, evals, proof, _ = ref[0]
        ot = oracle.transcript(b"open")
        exp = oracle.pcs_batch_open(maxsize, sets[0], [e for _, e in shape], points, evals, ot)
        assert exp.size == proof.size and (exp == proof).all()
        dpa.Basefold.batch_verify(maxsize, roots, [nv for nv, _ in shape], [not e for _, e in shape], points, evals, proof, dpa.Transcript(b"open"))
        assert eng.stats()["calls"] == 24  # (how many of them ran merged depends on how fast this interpreter submits: the first test pins the merging)
    finally:
        eng.close()


def test_blocking_calls_routed_to_an_engine_merge_across_threads_and_match_the_unrouted_calls(dev, oracle):
    """dp_ctx_route_to_engine (round 6): the BLOCKING seam calls of a context become submit + wait on an engine. Eight host threads make the same sumcheck / lookup /
    commit / fix_high / evaluate calls a seam-level host makes; every result equals the un-routed blocking call's, the transcripts end in the same state, and the
    engine's counters say that calls of different threads ran merged. (ctypes releases the GIL inside the calls: the threads really block concurrently.)"""
    import threading
    import deep_prove_amd as dpa
    rng = np.random.default_rng(909)
    pcs = dpa.Basefold(dev, 1 << 14)
    nv, n, T = 10, 1 << 10, 8
    jobs = []
    for j in range(T):
        tabs = [rand_base(rng, n) for _ in range(2)]
        cols = [rng.integers(0, 1 << 20, size=n, dtype=np.uint64) for _ in range(2)]
        jobs.append((tabs, cols, rand_point(rng, 1)[0], rand_point(rng, 1)[0], rand_point(rng, nv), rand_base(rng, 1 << 12)))

    def one(job, out, idx):
        tabs, cols, cc, csc, pt, poly = job
        t1, t2 = dpa.Transcript(b"routed"), dpa.Transcript(b"routed")
        m = [dpa.Mle.from_base(dev, tabs[0]), dpa.Mle.from_base(dev, tabs[1])]
        vp = dpa.VirtualPolynomial(nv)
        vp.add_mle_list([m[0], m[1]], (1, 0))
        pw, fin = dpa.prove_parallel(dev, vp, t1)
        lw = dpa.logup_batch_prove(dev, [dpa.Mle.from_base(dev, c) for c in cols], 2, cc, csc, t2)
        ev = m[0].evaluate(pt)
        com = pcs.commit(dpa.Mle.from_base(dev, poly))
        out[idx] = (pw, fin, t1.read_challenge(), lw, t2.read_challenge(), ev, np.array(com.root))

    ref = [None] * T
    for j in range(T):
        one(jobs[j], ref, j)
    eng = dpa.AsyncEngine(dev, max_in_flight=16, worker_arena_bytes=256 << 20)
    try:
        eng.route_blocking_calls()
        got = [None] * T
        th = [threading.Thread(target=one, args=(jobs[j], got, j)) for j in range(T)]
        for t in th:
            t.start()
        for t in th:
            t.join()
        for j in range(T):
            assert got[j] is not None, "a routed thread failed"
            for a, b in zip(ref[j], got[j]):
                assert np.array_equal(np.asarray(a), np.asarray(b))
        st = eng.stats()
        assert st["calls"] >= 4 * T and st["merged_calls"] >= 2, st  # (how many meet in one group is timing; that the merged path ran is not)
        eng.route_blocking_calls(False)
        again = [None]
        one(jobs[0], again, 0)  # detached: the plain blocking path again
        for a, b in zip(ref[0], again[0]):
            assert np.array_equal(np.asarray(a), np.asarray(b))
    finally:
        eng.close()
