"""ctypes access to oracle/liboracle.so — TEST INFRASTRUCTURE ONLY (never imported by the product package)."""
import ctypes as C
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
ORACLE_DIR = os.path.join(ROOT, "oracle")
u64p = C.POINTER(C.c_uint64)
i64p = C.POINTER(C.c_int64)
i32p = C.POINTER(C.c_int32)
_lib = None


def build():
    subprocess.check_call(["make", "-s", "-C", ORACLE_DIR])


def load():
    global _lib
    if _lib is None:
        so = os.path.join(ORACLE_DIR, "liboracle.so")
        srcs = [os.path.join(ORACLE_DIR, f) for f in os.listdir(ORACLE_DIR) if f.endswith((".hpp", ".cpp"))]
        if not os.path.exists(so) or any(os.path.getmtime(s) > os.path.getmtime(so) for s in srcs):
            build()
        lib = C.CDLL(so)
        lib.orc_last_error.restype = C.c_char_p
        lib.orc_transcript_new.restype = C.c_void_p
        lib.orc_transcript_new.argtypes = [C.c_char_p]
        lib.orc_transcript_free.argtypes = [C.c_void_p]
        lib.orc_free.argtypes = [C.c_void_p]
        _lib = Oracle(lib)
    return _lib


def _pt(pt):
    return np.array([w for e in pt for w in e], dtype=np.uint64)


class OracleTranscript:
    def __init__(self, lib, label=b"m2vec"):
        self.lib = lib
        self.h = C.c_void_p(lib.orc_transcript_new(label))

    def append_field_elements(self, e):
        a = np.ascontiguousarray(e, dtype=np.uint64)
        self.lib.orc_transcript_append_elements(self.h, a.ctypes.data_as(u64p), C.c_size_t(a.size))

    def append_message(self, m):
        self.lib.orc_transcript_append_message(self.h, m, C.c_size_t(len(m)))

    def get_and_append_challenge(self, label):
        out = (C.c_uint64 * 2)()
        self.lib.orc_transcript_challenge(self.h, label, out)
        return (int(out[0]), int(out[1]))

    def read_challenge(self):
        out = (C.c_uint64 * 2)()
        self.lib.orc_transcript_challenge(self.h, None, out)
        return (int(out[0]), int(out[1]))

    def __del__(self):
        try:
            self.lib.orc_transcript_free(self.h)
        except Exception:
            pass


class Oracle:
    def __init__(self, lib):
        self.lib = lib

    def set_gelu_files_lookup_claim(self, on):
        """False (the library's default): a GELU's prover files the claim the reference files (activation.rs:419-430); True: the claim its verifier checks"""
        self.lib.orc_set_gelu_files_lookup_claim(C.c_int(1 if on else 0))

    def _ok(self, rc):
        if rc != 0:
            raise RuntimeError("oracle: " + self.lib.orc_last_error().decode())

    def _take(self, ptr, n):
        out = np.ctypeslib.as_array(ptr, shape=(n,)).copy()
        self.lib.orc_free(ptr)
        return out

    def transcript(self, label=b"m2vec"):
        return OracleTranscript(self.lib, label)

    def selftest(self):
        return self.lib.orc_selftest()

    def rs_selftest(self, seed):
        return self.lib.orc_rs_selftest(C.c_uint64(seed))

    def rc_table(self):
        out = np.zeros(94, dtype=np.uint64)
        self.lib.orc_rc_table(out.ctypes.data_as(u64p))
        return out

    def permute(self, state):
        s = np.array(state, dtype=np.uint64)
        self.lib.orc_poseidon2_permute(s.ctypes.data_as(u64p))
        return s

    def trace_begin(self):
        """record the sponge traffic of every transcript the oracle creates on this thread until trace_take()"""
        self._ok(self.lib.orc_trace_begin())

    def trace_take(self):
        """-> uint64 array of (kind, value) pairs: kind 0 = absorbed, 1 = squeezed"""
        pw, pn = u64p(), C.c_size_t()
        self._ok(self.lib.orc_trace_take(C.byref(pw), C.byref(pn)))
        return self._take(pw, pn.value)

    def compress(self, x, y):
        a, b, o = (C.c_uint64 * 4)(*x), (C.c_uint64 * 4)(*y), (C.c_uint64 * 4)()
        self.lib.orc_compress(a, b, o)
        return [int(v) for v in o]

    def hash_or_noop(self, elems):
        w = np.ascontiguousarray(elems, dtype=np.uint64)
        o = (C.c_uint64 * 4)()
        self._ok(self.lib.orc_hash_or_noop(w.ctypes.data_as(u64p), C.c_size_t(w.size), o))
        return [int(v) for v in o]

    def eq_table(self, pt):
        p = _pt(pt)
        out = np.zeros(2 << len(pt), dtype=np.uint64)
        self._ok(self.lib.orc_eq_table(p.ctypes.data_as(u64p), C.c_uint32(len(pt)), out.ctypes.data_as(u64p)))
        return out

    def mle_eval(self, words, is_ext, pt):
        w = np.ascontiguousarray(words, dtype=np.uint64)
        p = _pt(pt)
        out = (C.c_uint64 * 2)()
        n = w.size // 2 if is_ext else w.size
        self._ok(self.lib.orc_mle_eval(w.ctypes.data_as(u64p), C.c_size_t(n), C.c_int(1 if is_ext else 0), p.ctypes.data_as(u64p), C.c_uint32(len(pt)), out))
        return (int(out[0]), int(out[1]))

    def fix_high(self, words, rows, cols, pt):
        w = np.ascontiguousarray(words, dtype=np.uint64)
        p = _pt(pt)
        out = np.zeros(2 * cols, dtype=np.uint64)
        self._ok(self.lib.orc_fix_high(w.ctypes.data_as(u64p), C.c_size_t(rows), C.c_size_t(cols), p.ctypes.data_as(u64p), out.ctypes.data_as(u64p)))
        return out

    def sumcheck_prove(self, nv, tables, is_ext, terms, transcript):
        """tables: list of uint64 arrays (2^k base words or 2 * 2^k ext words each, k <= nv); terms: list of (coeff, [idx])"""
        keep = [np.ascontiguousarray(t, dtype=np.uint64) for t in tables]
        tnv = np.array([(k.size // (2 if e else 1)).bit_length() - 1 for k, e in zip(keep, is_ext)], dtype=np.uint32)
        tp = (u64p * len(keep))(*[k.ctypes.data_as(u64p) for k in keep])
        ie = np.array([1 if e else 0 for e in is_ext], dtype=np.int32)
        deg = np.array([len(ix) for _, ix in terms], dtype=np.int32)
        tt = np.array([j for _, ix in terms for j in ix], dtype=np.int32)  # ragged: the terms' table lists back to back
        co = np.array([w for c, _ in terms for w in c], dtype=np.uint64)
        pw, pn = u64p(), C.c_size_t()
        finals = np.zeros(2 * len(keep), dtype=np.uint64)
        self._ok(self.lib.orc_sumcheck_prove(C.c_uint32(nv), tp, ie.ctypes.data_as(i32p), tnv.ctypes.data_as(C.POINTER(C.c_uint32)), C.c_int32(len(keep)), deg.ctypes.data_as(i32p),
                                             tt.ctypes.data_as(i32p), co.ctypes.data_as(u64p), C.c_int32(len(terms)), transcript.h,
                                             C.byref(pw), C.byref(pn), finals.ctypes.data_as(u64p)))
        return self._take(pw, pn.value), finals

    def logup_prove(self, columns, cpi, cc, csc, transcript, multiplicities=None):
        keep = [np.ascontiguousarray(c, dtype=np.uint64) for c in columns]
        cp = (u64p * len(keep))(*[k.ctypes.data_as(u64p) for k in keep])
        m = np.ascontiguousarray(multiplicities, dtype=np.uint64) if multiplicities is not None else None
        pw, pn = u64p(), C.c_size_t()
        self._ok(self.lib.orc_logup_prove(cp, C.c_int32(len(keep)), C.c_size_t(keep[0].size), C.c_int32(cpi),
                                          m.ctypes.data_as(u64p) if m is not None else None, (C.c_uint64 * 2)(*cc), (C.c_uint64 * 2)(*csc),
                                          transcript.h, C.byref(pw), C.byref(pn)))
        return self._take(pw, pn.value)

    def pcs_commit_root(self, max_poly_size, words, is_ext):
        w = np.ascontiguousarray(words, dtype=np.uint64)
        root = (C.c_uint64 * 4)()
        n = w.size // 2 if is_ext else w.size
        self._ok(self.lib.orc_pcs_commit_root(C.c_size_t(max_poly_size), w.ctypes.data_as(u64p), C.c_size_t(n), C.c_int(1 if is_ext else 0), root))
        return [int(x) for x in root]

    def pcs_open(self, max_poly_size, words, is_ext, point, transcript=None):
        w = np.ascontiguousarray(words, dtype=np.uint64)
        pt = _pt(point)
        pw, pn = u64p(), C.c_size_t()
        self._ok(self.lib.orc_pcs_open(C.c_size_t(max_poly_size), w.ctypes.data_as(u64p), C.c_size_t(w.size // 2 if is_ext else w.size), C.c_int(1 if is_ext else 0),
                                       pt.ctypes.data_as(u64p), transcript.h if transcript is not None else None, C.byref(pw), C.byref(pn)))
        return self._take(pw, pn.value)

    def pcs_batch_open(self, max_poly_size, polys, is_ext, points, evals, transcript):
        keep = [np.ascontiguousarray(p, dtype=np.uint64) for p in polys]
        pp = (u64p * len(keep))(*[k.ctypes.data_as(u64p) for k in keep])
        lens = (C.c_size_t * len(keep))(*[(k.size // 2 if e else k.size) for k, e in zip(keep, is_ext)])
        ie = np.array([1 if e else 0 for e in is_ext], dtype=np.int32)
        pf = np.concatenate([_pt(p) for p in points]).astype(np.uint64)
        ev = _pt(evals)
        pw, pn = u64p(), C.c_size_t()
        self._ok(self.lib.orc_pcs_batch_open(C.c_size_t(max_poly_size), pp, lens, ie.ctypes.data_as(i32p), C.c_int32(len(keep)), pf.ctypes.data_as(u64p),
                                             ev.ctypes.data_as(u64p), transcript.h, C.byref(pw), C.byref(pn)))
        return self._take(pw, pn.value)

    def pcs_batch_open_evals(self, max_poly_size, polys, is_ext, points, evals, transcript):
        """batch_open over a general Evaluation list: evals = [(poly index, point index, value)]"""
        keep = [np.ascontiguousarray(p, dtype=np.uint64) for p in polys]
        pp = (u64p * len(keep))(*[k.ctypes.data_as(u64p) for k in keep])
        lens = (C.c_size_t * len(keep))(*[(k.size // 2 if e else k.size) for k, e in zip(keep, is_ext)])
        ie = np.array([1 if e else 0 for e in is_ext], dtype=np.int32)
        pf = np.concatenate([_pt(p) for p in points]).astype(np.uint64)
        pl = np.array([len(p) for p in points], dtype=np.uint32)
        ep = np.array([e[0] for e in evals], dtype=np.uint32)
        eq = np.array([e[1] for e in evals], dtype=np.uint32)
        ev = _pt([e[2] for e in evals])
        u32p = C.POINTER(C.c_uint32)
        pw, pn = u64p(), C.c_size_t()
        self._ok(self.lib.orc_pcs_batch_open_evals(C.c_size_t(max_poly_size), pp, lens, ie.ctypes.data_as(i32p), C.c_int32(len(keep)), pf.ctypes.data_as(u64p), pl.ctypes.data_as(u32p),
                                                   C.c_int32(len(points)), ep.ctypes.data_as(u32p), eq.ctypes.data_as(u32p), ev.ctypes.data_as(u64p), C.c_int32(len(evals)), transcript.h,
                                                   C.byref(pw), C.byref(pn)))
        return self._take(pw, pn.value)

    def pcs_simple_batch_open(self, max_poly_size, polys, is_ext, point=None, transcript=None):
        """batch_commit + simple_batch_open of equally sized polynomials: (root, proof stream); point None: the root only"""
        keep = [np.ascontiguousarray(p, dtype=np.uint64) for p in polys]
        pp = (u64p * len(keep))(*[k.ctypes.data_as(u64p) for k in keep])
        n = keep[0].size // 2 if is_ext else keep[0].size
        root = (C.c_uint64 * 4)()
        pw, pn = u64p(), C.c_size_t()
        pt = _pt(point) if point is not None else None
        self._ok(self.lib.orc_pcs_simple_batch_open(C.c_size_t(max_poly_size), pp, C.c_size_t(n), C.c_int32(len(keep)), C.c_int(1 if is_ext else 0),
                                                    pt.ctypes.data_as(u64p) if pt is not None else None, transcript.h if transcript is not None else None, root,
                                                    C.byref(pw) if pt is not None else None, C.byref(pn) if pt is not None else None))
        return [int(x) for x in root], (self._take(pw, pn.value) if pt is not None else None)

    def model_setup(self, blob):
        b = np.ascontiguousarray(blob, dtype=np.int64)
        h = C.c_void_p()
        self._ok(self.lib.orc_model_setup(b.ctypes.data_as(i64p), C.c_size_t(b.size), C.byref(h)))
        return h

    def model_free(self, h):
        self.lib.orc_model_free(h)

    def model_prove(self, h, x):
        x = np.ascontiguousarray(x, dtype=np.int64)
        pw, pn = u64p(), C.c_size_t()
        out = np.zeros(1 << 20, dtype=np.int64)
        no = C.c_size_t(out.size)
        ms = C.c_double()
        self._ok(self.lib.orc_model_prove(h, x.ctypes.data_as(i64p), C.c_size_t(x.size), C.byref(pw), C.byref(pn), out.ctypes.data_as(i64p), C.byref(no), C.byref(ms)))
        return self._take(pw, pn.value), out[:no.value].copy(), ms.value

    def model_prove_many(self, h, x, threads, per_thread=1):
        """`threads` host threads x `per_thread` independent proofs; returns (wall ms, wrapping word-sum over all proofs)"""
        x = np.ascontiguousarray(x, dtype=np.int64)
        ms, dg = C.c_double(), C.c_uint64()
        self._ok(self.lib.orc_model_prove_many(h, x.ctypes.data_as(i64p), C.c_size_t(x.size), C.c_int32(threads), C.c_int32(per_thread), C.byref(ms), C.byref(dg)))
        return ms.value, int(dg.value)

    def model_prove_mt(self, h, x, threads):
        """ONE proof on `threads` cores (oracle/par.hpp); returns (wall ms, wrapping word-sum of the proof stream)"""
        x = np.ascontiguousarray(x, dtype=np.int64)
        ms, dg = C.c_double(), C.c_uint64()
        self._ok(self.lib.orc_model_prove_mt(h, x.ctypes.data_as(i64p), C.c_size_t(x.size), C.c_int32(threads), C.byref(ms), C.byref(dg)))
        return ms.value, int(dg.value)

    def bench_sumcheck(self, nv, k, seed):
        s = C.c_double()
        dg = (C.c_uint64 * 2)()
        self._ok(self.lib.orc_bench_sumcheck(C.c_uint32(nv), C.c_int32(k), C.c_uint64(seed), C.byref(s), dg))
        return s.value, (int(dg[0]), int(dg[1]))
