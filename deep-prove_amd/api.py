"""Host-side mirror of the reference's operator surface for the sumcheck / logup-GKR / Basefold hot path:

    reference (Rust)                                             here
    -----------------------------------------------------------  -----------------------------------------
    transcript::BasicTranscript                                  Transcript
    multilinear_extensions::DenseMultilinearExtension            Mle (device resident table)
    sumcheck::IOPProverState::prove_parallel(VirtualPolynomial)  VirtualPolynomial + prove_parallel
    zkml::lookup::logup_gkr::prover::batch_prove                 logup_batch_prove
    mpcs::PolynomialCommitmentScheme (Basefold RS/Poseidon)      Basefold.setup/commit/batch_open/batch_verify
    zkml::Context::generate / Prover::prove / verify             Context.generate / Prover.prove / verify

All O(n) work happens in libdeepprove_hip.so on the MI355X; field elements are canonical python ints / numpy uint64.
"""
import ctypes as C

import numpy as np

from . import _lib
from ._lib import check, u64p, i64p, i32p, u32p, vp

P = 0xFFFFFFFF00000001


def _u64(a):
    a = np.ascontiguousarray(a, dtype=np.uint64)
    return a, a.ctypes.data_as(u64p)


class _Malloced:
    """owner of a buffer malloc'ed by the library: numpy arrays made from it keep it alive through `.base`, dp_free runs
    when the last of them goes away. A proof is 5.9 MB (Dense-4M): copying it into a fresh numpy array cost 1.5 ms of
    single-threaded Python per proof — a third of a 192-proof batch."""

    def __init__(self, ptr, n):
        self._ptr = ptr
        self._free = _lib.load().dp_free
        self.__array_interface__ = {"data": (C.cast(ptr, C.c_void_p).value, False), "shape": (n,), "typestr": "<u8", "version": 3}

    def __del__(self):
        if self._ptr is not None:
            self._free(self._ptr)
            self._ptr = None


def _take(ptr, n):
    """wrap a malloc'ed uint64 buffer returned by the library as a numpy array WITHOUT copying; freed with the array"""
    if not n:
        _lib.load().dp_free(ptr)
        return np.zeros(0, dtype=np.uint64)
    return np.asarray(_Malloced(ptr, n))


class Device:
    """dp_ctx: one MI355X, one stream, one arena. Raises if no HIP device is present (no CPU fallback)."""

    def __init__(self, device_id=0):
        self._lib = _lib.load()
        h = vp()
        check(self._lib.dp_ctx_create(device_id, C.byref(h)))
        self.h = h
        self._contexts = []  # live Contexts (dp_model) generated on this device

    @property
    def name(self):
        return self._lib.dp_ctx_name(self.h).decode()

    def profile(self, on):
        check(self._lib.dp_profile_enable(self.h, 1 if on else 0))

    def probe_compress_rate(self, nodes=1 << 21, reps=5):
        """Poseidon2 compress() per second of the Merkle-layer kernel on a chip-filling layer (dp_probe_compress_rate)"""
        r = C.c_double()
        check(self._lib.dp_probe_compress_rate(self.h, nodes, reps, C.byref(r)))
        return r.value

    def profile_report(self):
        import json
        s = C.c_void_p()
        check(self._lib.dp_profile_report(self.h, C.byref(s)))
        try:
            return json.loads(C.string_at(s.value).decode())
        finally:
            self._lib.dp_free(s)  # the report is malloc'ed by the library

    def set_throughput_mode(self, on=True):
        """dp_ctx_set_throughput_mode: device-side Fiat-Shamir and the fused protocol kernels for this context's seam-level calls (what the
        workers of prove_batch run with); results are bit-identical to latency mode"""
        check(self._lib.dp_ctx_set_throughput_mode(self.h, 1 if on else 0))

    def close(self):
        """destroys the dp_ctx; models generated on it hold a dangling device afterwards, so they are freed first"""
        for c in list(self._contexts):
            c.free()
        if self.h:
            self._lib.dp_ctx_destroy(self.h)
            self.h = None


class Transcript:
    """transcript::BasicTranscript (transcript/src/basic.rs:8-54)"""

    def __init__(self, label=b"m2vec"):
        self._lib = _lib.load()
        self.h = vp(self._lib.dp_transcript_new(label))

    def append_field_elements(self, elems):
        a, p = _u64(elems)
        check(self._lib.dp_transcript_append_elements(self.h, p, a.size))

    def append_message(self, msg: bytes):
        check(self._lib.dp_transcript_append_message(self.h, msg, len(msg)))

    def append_usize(self, v: int):
        """append_message(&v.to_le_bytes()) — how the provers absorb num_vars / max_degree (sumcheck/src/prover.rs:507-511)"""
        self.append_message(int(v).to_bytes(8, "little"))

    def append_exts(self, exts):
        """extension elements (c0, c1) absorbed as two base elements each"""
        self.append_field_elements(np.array([w for e in exts for w in e], dtype=np.uint64))

    def get_and_append_challenge(self, label: bytes):
        out = (C.c_uint64 * 2)()
        check(self._lib.dp_transcript_challenge(self.h, label, out))
        return (int(out[0]), int(out[1]))

    def read_challenge(self):
        out = (C.c_uint64 * 2)()
        check(self._lib.dp_transcript_challenge(self.h, None, out))
        return (int(out[0]), int(out[1]))

    def __del__(self):
        try:
            if self.h:
                self._lib.dp_transcript_free(self.h)
                self.h = None
        except Exception:
            pass


def _point(pt):
    """list of (c0, c1) -> flat uint64"""
    return np.array([w for e in pt for w in e], dtype=np.uint64)


class Mle:
    """DenseMultilinearExtension resident in HBM (FieldType::Base or FieldType::Ext)."""

    def __init__(self, dev, handle):
        self.dev, self.h = dev, handle
        self._lib = _lib.load()

    @staticmethod
    def from_i64(dev, v):
        v = np.ascontiguousarray(v, dtype=np.int64)
        h = vp()
        check(_lib.load().dp_buf_from_i64(dev.h, v.ctypes.data_as(i64p), v.size, C.byref(h)))
        return Mle(dev, h)

    @staticmethod
    def from_base(dev, words):
        a, p = _u64(words)
        h = vp()
        check(_lib.load().dp_buf_upload(dev.h, p, a.size, 0, C.byref(h)))
        return Mle(dev, h)

    @staticmethod
    def from_ext(dev, words):
        a, p = _u64(words)
        h = vp()
        check(_lib.load().dp_buf_upload(dev.h, p, a.size // 2, 1, C.byref(h)))
        return Mle(dev, h)

    def __len__(self):
        return self._lib.dp_buf_len(self.h)

    @property
    def is_ext(self):
        return bool(self._lib.dp_buf_is_ext(self.h))

    @property
    def num_vars(self):
        return len(self).bit_length() - 1

    def to_numpy(self):
        out = np.zeros(len(self) * (2 if self.is_ext else 1), dtype=np.uint64)
        check(self._lib.dp_buf_download(self.dev.h, self.h, out.ctypes.data_as(u64p)))
        return out

    def evaluate(self, point):
        a, p = _u64(_point(point))
        out = (C.c_uint64 * 2)()
        check(self._lib.dp_mle_eval(self.dev.h, self.h, p, len(point), out))
        return (int(out[0]), int(out[1]))

    def fix_high_variables(self, rows, cols, point):
        """rows x cols base matrix, log2(rows)-coordinate point -> Mle of `cols` extension elements (K2)"""
        a, p = _u64(_point(point))
        h = vp()
        check(self._lib.dp_mle_fix_high(self.dev.h, self.h, rows, cols, p, C.byref(h)))
        return Mle(self.dev, h)

    def free(self):
        if self.h:
            self._lib.dp_buf_free(self.dev.h, self.h)
            self.h = None


def build_eq_x_r(dev, point):
    a, p = _u64(_point(point))
    h = vp()
    check(_lib.load().dp_eq_table(dev.h, p, len(point), C.byref(h)))
    return Mle(dev, h)


class VirtualPolynomial:
    """sum_i c_i * prod_j MLE  (multilinear_extensions/src/virtual_poly.rs:50-60)"""

    def __init__(self, num_vars):
        self.num_vars = num_vars
        self.tables = []
        self.terms = []  # (coeff (c0,c1), [table indices])

    def add_mle_list(self, mles, coeff=(1, 0)):
        idx = []
        for m in mles:
            for i, t in enumerate(self.tables):
                if t is m:
                    idx.append(i)
                    break
            else:
                self.tables.append(m)
                idx.append(len(self.tables) - 1)
        self.terms.append((coeff, idx))


def prove_parallel(dev, vpoly, transcript):
    """IOPProverState::prove_parallel: returns (proof_words, final_evaluations)"""
    lib = _lib.load()
    nt = len(vpoly.tables)
    tabs = (vp * nt)(*[t.h for t in vpoly.tables])
    deg = np.array([len(ix) for _, ix in vpoly.terms], dtype=np.int32)
    tt = np.array([j for _, ix in vpoly.terms for j in ix], dtype=np.int32)  # ragged: the terms' table lists back to back
    co = np.array([w for c, _ in vpoly.terms for w in c], dtype=np.uint64)
    pw, pn = u64p(), C.c_size_t()
    finals = np.zeros(2 * nt, dtype=np.uint64)
    check(lib.dp_sumcheck_prove(dev.h, vpoly.num_vars, tabs, nt, deg.ctypes.data_as(i32p), tt.ctypes.data_as(i32p),
                                co.ctypes.data_as(u64p), len(vpoly.terms), transcript.h, C.byref(pw), C.byref(pn),
                                finals.ctypes.data_as(u64p)))
    return _take(pw, pn.value), finals


def verify_sumcheck(claimed_sum, proof_words, num_vars, max_degree, transcript):
    """IOPVerifierState::verify (sumcheck/src/verifier.rs:12-168), host only: returns the SubClaim as
    (point [(c0, c1)] * num_vars, expected_evaluation (c0, c1)); raises DeepProveError(DP_ERR_VERIFY) on rejection"""
    pw = np.ascontiguousarray(proof_words, dtype=np.uint64)
    cs = (C.c_uint64 * 2)(*claimed_sum)
    pt = np.zeros(2 * max(num_vars, 1), dtype=np.uint64)
    ev = (C.c_uint64 * 2)()
    check(_lib.load().dp_sumcheck_verify(num_vars, max_degree, cs, pw.ctypes.data_as(u64p), pw.size, transcript.h,
                                         pt.ctypes.data_as(u64p), ev))
    return [(int(pt[2 * i]), int(pt[2 * i + 1])) for i in range(num_vars)], (int(ev[0]), int(ev[1]))


def verify_logup(proof_words, num_instances, constant_challenge, column_separation_challenge, transcript):
    """logup_gkr::verifier::verify_logup_proof (zkml/src/lookup/logup_gkr/verifier.rs:16-211), host only: returns
    (numerators, denominators, claims) with claims = [(point, eval)]; raises DeepProveError(DP_ERR_VERIFY) on rejection"""
    pw = np.ascontiguousarray(proof_words, dtype=np.uint64)
    cc = (C.c_uint64 * 2)(*constant_challenge)
    cs = (C.c_uint64 * 2)(*column_separation_challenge)
    num = np.zeros(2 * num_instances, dtype=np.uint64)
    den = np.zeros(2 * num_instances, dtype=np.uint64)
    cw, cn = u64p(), C.c_size_t()
    check(_lib.load().dp_logup_verify(pw.ctypes.data_as(u64p), pw.size, num_instances, cc, cs, transcript.h,
                                      num.ctypes.data_as(u64p), den.ctypes.data_as(u64p), C.byref(cw), C.byref(cn)))
    w = _take(cw, cn.value)
    pos, claims = 1, []
    for _ in range(int(w[0])):
        k = int(w[pos]); pos += 1
        point = [(int(w[pos + 2 * i]), int(w[pos + 2 * i + 1])) for i in range(k)]
        pos += 2 * k
        claims.append((point, (int(w[pos]), int(w[pos + 1]))))
        pos += 2
    pair = lambda a: [(int(a[2 * i]), int(a[2 * i + 1])) for i in range(num_instances)]  # noqa: E731
    return pair(num), pair(den), claims


def logup_batch_prove(dev, columns, columns_per_instance, constant_challenge, column_separation_challenge, transcript,
                      multiplicities=None):
    lib = _lib.load()
    cols = (vp * len(columns))(*[c.h for c in columns])
    cc = (C.c_uint64 * 2)(*constant_challenge)
    cs = (C.c_uint64 * 2)(*column_separation_challenge)
    pw, pn = u64p(), C.c_size_t()
    check(lib.dp_logup_prove(dev.h, cols, len(columns), columns_per_instance,
                             multiplicities.h if multiplicities is not None else None, cc, cs, transcript.h,
                             C.byref(pw), C.byref(pn)))
    return _take(pw, pn.value)


class Ticket:
    """a submitted seam call (dp_ticket): poll() -> 0 running / 1 done (raises DeepProveError if it failed); the results once done"""

    def __init__(self, handle, keep):
        self.h, self._keep = vp(handle), keep  # `keep`: the argument arrays and handles the call reads until it completes

    def poll(self):
        s = _lib.load().dp_poll(self.h)
        if s < 0:
            check(s)
        return s

    def wait(self):
        check(_lib.load().dp_wait(self.h))
        return self

    def words(self, which=0):
        pw, pn = u64p(), C.c_size_t()
        check(_lib.load().dp_ticket_words(self.h, which, C.byref(pw), C.byref(pn)))
        return _take(pw, pn.value)

    def values(self, n):
        out = np.zeros(n, dtype=np.uint64)
        check(_lib.load().dp_ticket_values(self.h, out.ctypes.data_as(u64p), n))
        return out

    def commitment(self, dev, poly):
        h, root = vp(), (C.c_uint64 * 4)()
        check(_lib.load().dp_ticket_commit(self.h, C.byref(h), root))
        return Commitment(dev, h, [int(v) for v in root], poly)

    def table(self, dev):
        """the device table of a completed commit_host / fix_high ticket (dp_ticket_buf)"""
        h = vp()
        check(_lib.load().dp_ticket_buf(self.h, C.byref(h)))
        return Mle(dev, h)

    def free(self):
        if self.h:
            check(_lib.load().dp_ticket_free(self.h))
            self.h, self._keep = None, None


class AsyncEngine:
    """dp_async: submit / poll forms of the seam calls — one host thread keeps hundreds of calls in flight; calls of identical shape queued together are
    proved in lock step with merged launches (include/deep_prove_hip.h, "asynchronous seam calls"). Create it after Basefold(dev, ..) (dp_pcs_setup)."""

    def __init__(self, dev, max_in_flight=64, worker_arena_bytes=0):
        self.dev, self.h = dev, vp()
        check(_lib.load().dp_async_create(dev.h, max_in_flight, worker_arena_bytes, C.byref(self.h)))

    def close(self):
        if self.h:
            self.route_blocking_calls(False)
            check(_lib.load().dp_async_destroy(self.h))
            self.h = None

    def route_blocking_calls(self, on=True):
        """dp_ctx_route_to_engine: the BLOCKING seam calls of the device's context (prove_parallel, logup_batch_prove, Basefold.commit / batch_open,
        fix_high, evaluate) become submit + wait on this engine — calls of one shape made by other threads at the same moment are merged"""
        check(_lib.load().dp_ctx_route_to_engine(self.dev.h, self.h if on else None))

    def stats(self):
        a, b, c, d = C.c_size_t(), C.c_size_t(), C.c_size_t(), C.c_size_t()
        check(_lib.load().dp_async_stats(self.h, C.byref(a), C.byref(b), C.byref(c), C.byref(d)))
        return {"calls": a.value, "groups": b.value, "merged_calls": c.value, "workers": d.value}

    def prove_parallel(self, vpoly, transcript):
        nt = len(vpoly.tables)
        tabs = (vp * nt)(*[t.h for t in vpoly.tables])
        deg = np.array([len(ix) for _, ix in vpoly.terms], dtype=np.int32)
        tt = np.array([j for _, ix in vpoly.terms for j in ix], dtype=np.int32)
        co = np.array([w for c, _ in vpoly.terms for w in c], dtype=np.uint64)
        t = vp()
        check(_lib.load().dp_sumcheck_prove_submit(self.h, vpoly.num_vars, tabs, nt, deg.ctypes.data_as(i32p), tt.ctypes.data_as(i32p), co.ctypes.data_as(u64p),
                                                   len(vpoly.terms), transcript.h, C.byref(t)))
        return Ticket(t.value, (vpoly, transcript))

    def logup_batch_prove(self, columns, columns_per_instance, constant_challenge, column_separation_challenge, transcript, multiplicities=None):
        cols = (vp * len(columns))(*[c.h for c in columns])
        cc = (C.c_uint64 * 2)(*constant_challenge)
        cs = (C.c_uint64 * 2)(*column_separation_challenge)
        t = vp()
        check(_lib.load().dp_logup_prove_submit(self.h, cols, len(columns), columns_per_instance, multiplicities.h if multiplicities is not None else None, cc, cs, transcript.h, C.byref(t)))
        return Ticket(t.value, (columns, multiplicities, transcript))

    def commit(self, poly):
        t = vp()
        check(_lib.load().dp_pcs_commit_submit(self.h, poly.h, C.byref(t)))
        return Ticket(t.value, (poly,))

    def commit_host(self, words, is_ext=False):
        """PCS::commit(&poly) with the polynomial on the host: upload + commit in one ticket; Ticket.table(dev) then Ticket.commitment(dev, table)"""
        w = np.ascontiguousarray(words, dtype=np.uint64)
        t = vp()
        check(_lib.load().dp_pcs_commit_host_submit(self.h, w.ctypes.data_as(u64p), w.size // (2 if is_ext else 1), 1 if is_ext else 0, C.byref(t)))
        return Ticket(t.value, None)

    def batch_open(self, comms, points, evals, transcript):
        hs = (vp * len(comms))(*[c.h for c in comms])
        pf = np.concatenate([_point(p) for p in points]).astype(np.uint64)
        ev = _point(evals)
        t = vp()
        check(_lib.load().dp_pcs_batch_open_submit(self.h, hs, len(comms), pf.ctypes.data_as(u64p), ev.ctypes.data_as(u64p), transcript.h, C.byref(t)))
        return Ticket(t.value, (comms, transcript))


class Commitment:
    def __init__(self, dev, handle, root, poly):
        self.dev, self.h, self.root, self.poly = dev, handle, root, poly

    @property
    def num_vars(self):
        return self.poly.num_vars

    def free(self):
        if self.h:
            _lib.load().dp_pcs_commit_free(self.dev.h, self.h)
            self.h = None


def _eval_lists(points, evals):
    """flat arrays of an Evaluation list: points, their lengths, (poly, point) indices, values"""
    pf = np.concatenate([_point(p) for p in points]).astype(np.uint64)
    pl = np.array([len(p) for p in points], dtype=np.uint32)
    ep = np.array([e[0] for e in evals], dtype=np.uint32)
    eq = np.array([e[1] for e in evals], dtype=np.uint32)
    ev = _point([e[2] for e in evals])
    return pf, pl, ep, eq, ev


class BatchCommitment:
    """BasefoldCommitmentWithWitness of several polynomials behind one root (mpcs/src/basefold.rs:356-446)"""

    def __init__(self, dev, handle, root, polys):
        self.dev, self.h, self.root, self.polys = dev, handle, root, polys

    @property
    def num_vars(self):
        return self.polys[0].num_vars

    def free(self):
        if self.h:
            _lib.load().dp_pcs_batch_commit_free(self.dev.h, self.h)
            self.h = None


class Basefold:
    """mpcs::PolynomialCommitmentScheme for Basefold<GoldilocksExt2, BasefoldRSParams<PoseidonHasher>>"""

    def __init__(self, dev, max_poly_size):
        self.dev, self.max_poly_size = dev, max_poly_size
        check(_lib.load().dp_pcs_setup(dev.h, max_poly_size))

    def commit(self, poly):
        h = vp()
        root = (C.c_uint64 * 4)()
        check(_lib.load().dp_pcs_commit(self.dev.h, poly.h, C.byref(h), root))
        return Commitment(self.dev, h, [int(x) for x in root], poly)

    def open(self, comm, point, eval_, transcript=None):
        """PCS::open (mpcs/src/basefold.rs:466-544): one polynomial at one point; <= 7 variables: the trivial proof (no transcript)"""
        pt, ev = _point(point), _point([eval_])
        pw, pn = u64p(), C.c_size_t()
        check(_lib.load().dp_pcs_open(self.dev.h, comm.h, pt.ctypes.data_as(u64p), len(point), ev.ctypes.data_as(u64p),
                                      transcript.h if transcript is not None else None, C.byref(pw), C.byref(pn)))
        return _take(pw, pn.value)

    @staticmethod
    def verify(max_poly_size, root, num_vars, is_base, point, eval_, proof_words, transcript=None):
        """PCS::verify (mpcs/src/basefold.rs:863-962). Host only. Raises DeepProveError(DP_ERR_VERIFY) on rejection."""
        r = np.array(root, dtype=np.uint64)
        pt, ev = _point(point), _point([eval_])
        pw = np.ascontiguousarray(proof_words, dtype=np.uint64)
        check(_lib.load().dp_pcs_verify(max_poly_size, r.ctypes.data_as(u64p), num_vars, 1 if is_base else 0, pt.ctypes.data_as(u64p),
                                        ev.ctypes.data_as(u64p), pw.ctypes.data_as(u64p), pw.size, transcript.h if transcript is not None else None))

    def batch_open_evals(self, comms, points, evals, transcript):
        """PCS::batch_open over a general Evaluation list (mpcs/src/basefold.rs:546-770): evals = [(poly index, point index, value)]"""
        hs = (vp * len(comms))(*[c.h for c in comms])
        pf, pl, ep, eq, ev = _eval_lists(points, evals)
        pw, pn = u64p(), C.c_size_t()
        check(_lib.load().dp_pcs_batch_open_evals(self.dev.h, hs, len(comms), pf.ctypes.data_as(u64p), pl.ctypes.data_as(u32p), len(points), ep.ctypes.data_as(u32p),
                                                  eq.ctypes.data_as(u32p), ev.ctypes.data_as(u64p), len(evals), transcript.h, C.byref(pw), C.byref(pn)))
        return _take(pw, pn.value)

    @staticmethod
    def batch_verify_evals(max_poly_size, roots, num_vars, is_base, points, evals, proof_words, transcript):
        """PCS::batch_verify over a general Evaluation list (mpcs/src/basefold.rs:964-1098). Host only."""
        r = np.array([w for root in roots for w in root], dtype=np.uint64)
        nv = np.array(num_vars, dtype=np.uint32)
        ib = np.array([1 if b else 0 for b in is_base], dtype=np.int32)
        pf, pl, ep, eq, ev = _eval_lists(points, evals)
        pw = np.ascontiguousarray(proof_words, dtype=np.uint64)
        check(_lib.load().dp_pcs_batch_verify_evals(max_poly_size, r.ctypes.data_as(u64p), nv.ctypes.data_as(u32p), ib.ctypes.data_as(i32p), len(roots), pf.ctypes.data_as(u64p),
                                                    pl.ctypes.data_as(u32p), len(points), ep.ctypes.data_as(u32p), eq.ctypes.data_as(u32p), ev.ctypes.data_as(u64p), len(evals),
                                                    pw.ctypes.data_as(u64p), pw.size, transcript.h))

    def batch_commit(self, polys):
        """PCS::batch_commit (mpcs/src/basefold.rs:356-446): 1..32 polynomials of one size and one field behind ONE Merkle root"""
        hs = (vp * len(polys))(*[p.h for p in polys])
        h = vp()
        root = (C.c_uint64 * 4)()
        check(_lib.load().dp_pcs_batch_commit(self.dev.h, hs, len(polys), C.byref(h), root))
        return BatchCommitment(self.dev, h, [int(x) for x in root], list(polys))

    def simple_batch_open(self, comm, point, transcript=None):
        """PCS::simple_batch_open (mpcs/src/basefold.rs:777-861): every polynomial of a batch commitment at one point"""
        pt = _point(point)
        pw, pn = u64p(), C.c_size_t()
        check(_lib.load().dp_pcs_simple_batch_open(self.dev.h, comm.h, pt.ctypes.data_as(u64p), len(point), transcript.h if transcript is not None else None,
                                                   C.byref(pw), C.byref(pn)))
        return _take(pw, pn.value)

    @staticmethod
    def simple_batch_verify(max_poly_size, root, num_vars, is_base, point, evals, proof_words, transcript=None):
        """PCS::simple_batch_verify (mpcs/src/basefold.rs:1100-1203). Host only. Raises DeepProveError(DP_ERR_VERIFY) on rejection."""
        r = np.array(root, dtype=np.uint64)
        pt, ev = _point(point), _point(evals)
        pw = np.ascontiguousarray(proof_words, dtype=np.uint64)
        check(_lib.load().dp_pcs_simple_batch_verify(max_poly_size, r.ctypes.data_as(u64p), num_vars, 1 if is_base else 0, pt.ctypes.data_as(u64p),
                                                     ev.ctypes.data_as(u64p), len(evals), pw.ctypes.data_as(u64p), pw.size,
                                                     transcript.h if transcript is not None else None))

    def batch_open(self, comms, points, evals, transcript):
        lib = _lib.load()
        hs = (vp * len(comms))(*[c.h for c in comms])
        pf = np.concatenate([_point(p) for p in points]).astype(np.uint64)
        ev = _point(evals)
        pw, pn = u64p(), C.c_size_t()
        check(lib.dp_pcs_batch_open(self.dev.h, hs, len(comms), pf.ctypes.data_as(u64p), ev.ctypes.data_as(u64p),
                                    transcript.h, C.byref(pw), C.byref(pn)))
        return _take(pw, pn.value)

    @staticmethod
    def batch_verify(max_poly_size, roots, num_vars, is_base, points, evals, proof_words, transcript):
        lib = _lib.load()
        r = np.array([w for root in roots for w in root], dtype=np.uint64)
        nv = np.array(num_vars, dtype=np.uint32)
        ib = np.array([1 if b else 0 for b in is_base], dtype=np.int32)
        pf = np.concatenate([_point(p) for p in points]).astype(np.uint64)
        ev = _point(evals)
        pw = np.ascontiguousarray(proof_words, dtype=np.uint64)
        check(lib.dp_pcs_batch_verify(max_poly_size, r.ctypes.data_as(u64p), nv.ctypes.data_as(u32p),
                                      ib.ctypes.data_as(i32p), len(roots), pf.ctypes.data_as(u64p),
                                      ev.ctypes.data_as(u64p), pw.ctypes.data_as(u64p), pw.size, transcript.h))


class Context:
    """zkml::Context (zkml/src/iop/context.rs:37-52): model commitments + PCS params + lookup tables, device resident."""

    def __init__(self, dev, handle, model_blob):
        self.dev, self.h, self.model_blob = dev, handle, model_blob
        dev._contexts.append(self)

    @staticmethod
    def generate(dev, model_blob):
        b = np.ascontiguousarray(model_blob, dtype=np.int64)
        h = vp()
        check(_lib.load().dp_model_setup(dev.h, b.ctypes.data_as(i64p), b.size, C.byref(h)))
        return Context(dev, h, b)

    def verifier_blob(self):
        pw, pn = u64p(), C.c_size_t()
        check(_lib.load().dp_model_verifier_blob(self.h, C.byref(pw), C.byref(pn)))
        return _take(pw, pn.value)

    def free(self):
        if self.h:
            _lib.load().dp_model_free(self.h)
            self.h = None
            if self in self.dev._contexts:
                self.dev._contexts.remove(self)


class Prover:
    """zkml::Prover (zkml/src/iop/prover.rs:40-58). prove() = Model::run on the host + Prover::prove on the device."""

    def __init__(self, ctx):
        self.ctx = ctx
        self.last_prove_ms = None
        self._nout = None

    def prove(self, input_i64):
        x = np.ascontiguousarray(input_i64, dtype=np.int64)
        pw, pn = u64p(), C.c_size_t()
        out = np.empty(self.output_len(), dtype=np.int64)
        no = C.c_size_t(out.size)
        ms = C.c_double()
        check(_lib.load().dp_model_prove(self.ctx.h, x.ctypes.data_as(i64p), x.size, C.byref(pw), C.byref(pn),
                                         out.ctypes.data_as(i64p), C.byref(no), C.byref(ms)))
        self.last_prove_ms = ms.value
        return _take(pw, pn.value), out[:no.value].copy()


    def prove_batch(self, inputs_i64, concurrency):
        """independent proofs with up to `concurrency` in flight on this GPU; returns ([proof_words], outputs[nproofs, nout], wall_ms)"""
        x = np.ascontiguousarray(inputs_i64, dtype=np.int64)
        nproofs, ninput = x.shape
        lib = _lib.load()
        pws = (u64p * nproofs)()
        pns = (C.c_size_t * nproofs)()
        cap = self.output_len()
        outs = np.empty((nproofs, cap), dtype=np.int64)
        no = C.c_size_t(0)
        ms = C.c_double()
        check(lib.dp_model_prove_batch(self.ctx.h, x.ctypes.data_as(i64p), nproofs, ninput, concurrency, pws, pns,
                                       outs.ctypes.data_as(i64p), cap, C.byref(no), C.byref(ms)))
        proofs = [_take(pws[i], pns[i]) for i in range(nproofs)]
        return proofs, outs[:, :no.value].copy(), ms.value


    def output_len(self):
        """length of the model's output tensor (dp_model_output_len)"""
        if self._nout is None:
            n = C.c_size_t(0)
            check(_lib.load().dp_model_output_len(self.ctx.h, C.byref(n)))
            self._nout = int(n.value)
        return self._nout

    def in_flight(self):
        """proofs the last prove_batch kept in flight (the `concurrency` asked, cut to what fits in the free HBM)"""
        n = C.c_size_t(0)
        check(_lib.load().dp_model_in_flight(self.ctx.h, C.byref(n)))
        return int(n.value)


def host_cpu_budget():
    """CPUs this process may use (cgroup quota, else hardware threads): what the number of proving host threads follows"""
    return float(_lib.load().dp_host_cpu_budget())


def infer_host(model_blob, input_i64):
    """dp_model_infer_host: the quantised inference of a model blob on the host (Model::run); no device involved"""
    b = np.ascontiguousarray(model_blob, dtype=np.int64)
    x = np.ascontiguousarray(input_i64, dtype=np.int64)
    out = np.empty(max(1 << 16, 4 * x.size), dtype=np.int64)
    n = C.c_size_t(out.size)
    rc = _lib.load().dp_model_infer_host(b.ctypes.data_as(i64p), b.size, x.ctypes.data_as(i64p), x.size, out.ctypes.data_as(i64p), C.byref(n))
    check(rc)
    return out[:n.value].copy()


def verify_batch(verifier_blob, proofs, inputs_i64, outputs_i64, dev=None, threads=0):
    """dp_verify_batch: verdict per proof (0 accepted, -5 rejected, -1 malformed) and the wall time in ms. With `dev` the Merkle
    paths are authenticated on the GPU (one launch per proof), the protocol checks run on host threads either way."""
    vb = np.ascontiguousarray(verifier_blob, dtype=np.uint64)
    keep = [np.ascontiguousarray(p, dtype=np.uint64) for p in proofs]
    n = len(keep)
    pws = (u64p * n)(*[k.ctypes.data_as(u64p) for k in keep])
    pns = (C.c_size_t * n)(*[k.size for k in keep])
    x = np.ascontiguousarray(inputs_i64, dtype=np.int64).reshape(n, -1)
    y = np.ascontiguousarray(outputs_i64, dtype=np.int64).reshape(n, -1)
    res = np.zeros(n, dtype=np.int32)
    ms = C.c_double()
    check(_lib.load().dp_verify_batch(dev.h if dev is not None else None, vb.ctypes.data_as(u64p), vb.size, pws, pns, x.ctypes.data_as(i64p), x.shape[1],
                                      y.ctypes.data_as(i64p), y.shape[1], n, threads, res.ctypes.data_as(i32p), C.byref(ms)))
    return res, ms.value


def verify(verifier_blob, proof_words, input_i64, output_i64):
    """zkml::verify (zkml/src/iop/verifier.rs:306-318). Host only. Raises DeepProveError(DP_ERR_VERIFY) on rejection."""
    vb = np.ascontiguousarray(verifier_blob, dtype=np.uint64)
    pw = np.ascontiguousarray(proof_words, dtype=np.uint64)
    x = np.ascontiguousarray(input_i64, dtype=np.int64)
    y = np.ascontiguousarray(output_i64, dtype=np.int64)
    check(_lib.load().dp_verify(vb.ctypes.data_as(u64p), vb.size, pw.ctypes.data_as(u64p), pw.size,
                                x.ctypes.data_as(i64p), x.size, y.ctypes.data_as(i64p), y.size))
