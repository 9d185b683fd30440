"""One sumcheck sharded over several devices (BASELINE config 5).

The partition is the reference's thread-sharded prover `IOPProverState::prove_batch_polys`
(sumcheck/src/prover.rs:37-321, merge step sumcheck/src/util.rs:215-243): worker g of W = 2^k owns the contiguous slice
[g N/W, (g+1) N/W) of every table, i.e. the top k variables select the worker. Variables are bound low to high, so the
first nv - k rounds are local: every worker computes the round sums of its slice, the shares are exchanged (an all-gather
of (d+1) extension elements per term: a few hundred bytes, pure latency) and added mod p, every worker runs the same
Fiat-Shamir transcript on the total and folds its slice with the same challenge. After nv - k rounds each worker holds one
value per table; those are all-gathered into tables of W entries and the last k rounds run (identically) on every worker.
The transcript sequence is that of the unsharded prover, so the proof is bit-identical to `prove_parallel`.

The exchange is `torch.distributed.all_gather` (RCCL over xGMI when the process group is "nccl": the shares travel as
int64 device tensors; "gloo" on CPU). A mod-p sum is not an RCCL reduction (canonical u64 words would overflow / not
reduce), hence gather + local modular add.
"""
import ctypes as C

import numpy as np

from . import _lib
from ._lib import check, vp

P = 0xFFFFFFFF00000001
W7 = 7  # X^2 = 7


# ---- degree-2 extension arithmetic on (c0, c1) tuples of Python ints (per round a handful of elements)
def e_add(a, b):
    return ((a[0] + b[0]) % P, (a[1] + b[1]) % P)


def e_sub(a, b):
    return ((a[0] - b[0]) % P, (a[1] - b[1]) % P)


def e_mul(a, b):
    return ((a[0] * b[0] + W7 * a[1] * b[1]) % P, (a[0] * b[1] + a[1] * b[0]) % P)


def e_scale(a, s):
    return (a[0] * s % P, a[1] * s % P)


def extrapolate(evals, at):
    """value at the integer `at` of the polynomial through (i, evals[i]), i = 0..k (sumcheck/src/util.rs:101-136)"""
    k = len(evals) - 1
    res = (0, 0)
    for i in range(k + 1):
        num, den = 1, 1
        for j in range(k + 1):
            if j != i:
                num = num * (at - j) % P
                den = den * (i - j) % P
        res = e_add(res, e_scale(evals[i], num * pow(den, P - 2, P) % P))
    return res


class HipShard:
    """the local slice of a virtual polynomial on one MI355X: dp_sc_session_* of include/deep_prove_hip.h"""

    def __init__(self, dev, nv_local, tables, terms):
        """tables: list of api.Mle with 2^nv_local entries; terms: list of (coeff (c0,c1), [table indices])"""
        self.lib = _lib.load()
        self.ntables = len(tables)
        self.terms = terms
        tabs = (vp * len(tables))(*[t.h for t in tables])
        deg = np.array([len(ix) for _, ix in terms], dtype=np.int32)
        tt = np.array([j for _, ix in terms for j in ix], dtype=np.int32)  # ragged: the terms' table lists back to back
        self.h = vp()
        check(self.lib.dp_sc_session_new(dev.h, nv_local, tabs, len(tables), deg.ctypes.data_as(_lib.i32p), tt.ctypes.data_as(_lib.i32p), len(terms), C.byref(self.h)))
        self.nraw = int(sum(len(ix) + 1 for _, ix in terms))

    def round(self, r_prev):
        raw = np.zeros(2 * self.nraw, dtype=np.uint64)
        rp = None
        if r_prev is not None:
            rp = np.array(r_prev, dtype=np.uint64).ctypes.data_as(_lib.u64p)
        check(self.lib.dp_sc_session_round(self.h, rp, raw.ctypes.data_as(_lib.u64p), None))
        return raw

    def finish(self, r_last):
        fin = np.zeros(2 * self.ntables, dtype=np.uint64)
        check(self.lib.dp_sc_session_finish(self.h, np.array(r_last, dtype=np.uint64).ctypes.data_as(_lib.u64p), fin.ctypes.data_as(_lib.u64p)))
        return fin

    def close(self):
        if self.h:
            self.lib.dp_sc_session_free(self.h)
            self.h = vp()


class TorchExchange:
    """all-gather of a small uint64 vector over a torch.distributed process group (RCCL for "nccl", gloo on CPU)"""

    def __init__(self, group=None, device=None):
        import torch
        import torch.distributed as dist
        self.torch, self.dist, self.group = torch, dist, group
        self.world = dist.get_world_size(group)
        self.rank = dist.get_rank(group)
        self.device = device if device is not None else ("cuda" if dist.get_backend(group) == "nccl" else "cpu")

    def all_gather(self, words):
        t = self.torch.from_numpy(words.view(np.int64).copy()).to(self.device)
        out = [self.torch.empty_like(t) for _ in range(self.world)]
        self.dist.all_gather(out, t, group=self.group)
        return [o.cpu().numpy().view(np.uint64) for o in out]


class LocalExchange:
    """W shards driven from one process (tests, single-GPU emulation): `all_gather` is fed by the driver loop"""

    def __init__(self, world):
        self.world, self.rank = world, 0


def _message(terms, raw_total, max_degree):
    """raw per-term sums -> the round message (max_degree + 1 evaluations), as sumcheck_prove does on the host"""
    msg = [(0, 0)] * (max_degree + 1)
    off = 0
    for coeff, ix in terms:
        k = len(ix)
        s = [e_mul((int(raw_total[2 * (off + j)]), int(raw_total[2 * (off + j) + 1])), coeff) for j in range(k + 1)]
        off += k + 1
        for j in range(max_degree + 1):
            msg[j] = e_add(msg[j], s[j] if j <= k else extrapolate(s, j))
    return msg


def _add_shares(shares):
    tot = [0] * len(shares[0])
    for sh in shares:
        for i, v in enumerate(sh):
            tot[i] = (tot[i] + int(v)) % P
    return tot


def prove_sharded(shards, exchange, nv, terms, transcript, make_small_shard):
    """The sharded prover. `shards`: the local shard objects of THIS process (one per device it drives: one with
    torch.distributed, all W with LocalExchange), each with round(r_prev) / finish(r_last) over 2^(nv - k) entries.
    `make_small_shard(tables_words)`: builds a shard over the gathered W-entry extension tables for the last k rounds.
    Returns (proof_words, finals) in the layout of api.prove_parallel."""
    world = exchange.world
    k = world.bit_length() - 1
    assert 1 << k == world and nv > k
    nv_local = nv - k
    max_degree = max(len(ix) for _, ix in terms)
    transcript.append_usize(nv)
    transcript.append_usize(max_degree)
    point, rounds = [], []

    def gather(per_shard):
        if isinstance(exchange, LocalExchange):
            return per_shard
        assert len(per_shard) == 1
        return exchange.all_gather(per_shard[0])

    r = None
    for _ in range(nv_local):
        raw_total = _add_shares(gather([s.round(r) for s in shards]))
        msg = _message(terms, raw_total, max_degree)
        transcript.append_exts(msg)
        rounds.append(msg)
        r = transcript.get_and_append_challenge(b"Internal round")
        point.append(r)
    finals_local = gather([s.finish(r) for s in shards])  # W arrays of 2 * ntables words
    for s in shards:
        s.close()  # the slices are consumed; their device context is free for the stage-2 shard
    ntables = len(finals_local[0]) // 2
    if k == 0:
        finals = [(int(finals_local[0][2 * j]), int(finals_local[0][2 * j + 1])) for j in range(ntables)]
    else:
        # stage 2 (merge_sumcheck_polys, util.rs:215-243): table j has one entry per worker, in worker order
        small = make_small_shard([np.array([w for g in range(world) for w in finals_local[g][2 * j:2 * j + 2]], dtype=np.uint64) for j in range(ntables)])
        r = None
        for _ in range(k):
            msg = _message(terms, _add_shares([small.round(r)]), max_degree)
            transcript.append_exts(msg)
            rounds.append(msg)
            r = transcript.get_and_append_challenge(b"Internal round")
            point.append(r)
        f = small.finish(r)
        small.close()
        finals = [(int(f[2 * j]), int(f[2 * j + 1])) for j in range(ntables)]
    # IOPProof stream {point: len, ext...; rounds: count, (len, ext...)...} (csrc/proof.h Writer::iop)
    words = [len(point)] + [w for p in point for w in p] + [len(rounds)]
    for m in rounds:
        words += [len(m)] + [w for e in m for w in e]
    return np.array(words, dtype=np.uint64), np.array([w for e in finals for w in e], dtype=np.uint64)


# ---------------------------------------------------------------- the in-library loop (csrc/sharded.h): this module is only the driver
def _pack_terms(terms):
    deg = np.array([len(ix) for _, ix in terms], dtype=np.int32)
    tt = np.array([j for _, ix in terms for j in ix], dtype=np.int32)
    co = np.array([w for c, _ in terms for w in c], dtype=np.uint64)
    return deg, tt, co


class RcclGroup:
    """dp_dist: this rank's RCCL communicator inside the library. The 128-byte unique id travels over the torch.distributed
    control plane (an object broadcast: it works on a gloo group too); the data path is ncclAllGather on device buffers."""

    def __init__(self, dev, group=None):
        import torch.distributed as dist
        lib = _lib.load()
        self.rank, self.world = dist.get_rank(group), dist.get_world_size(group)
        ident = (C.c_uint8 * 128)()
        if self.rank == 0:
            check(lib.dp_dist_unique_id(ident))
        box = [bytes(ident)]
        dist.broadcast_object_list(box, src=0, group=group)
        ident = (C.c_uint8 * 128)(*box[0])
        self.h = vp()
        check(lib.dp_dist_init(dev.h, ident, self.rank, self.world, C.byref(self.h)))

    def close(self):
        if self.h:
            _lib.load().dp_dist_free(self.h)
            self.h = vp()


def prove_sharded_in_library(dev, group, nv, local_tables, terms, transcript):
    """dp_sumcheck_prove_sharded: the whole round loop (local round sums, ncclAllGather, mod-p sum, host sponge, stage 2) in C++;
    `group`: an RcclGroup or None (world of one). Returns (proof_words, finals) as api.prove_parallel."""
    from .api import _take
    lib = _lib.load()
    nt = len(local_tables)
    tabs = (vp * nt)(*[t.h for t in local_tables])
    deg, tt, co = _pack_terms(terms)
    pw, pn = _lib.u64p(), C.c_size_t()
    finals = np.zeros(2 * nt, dtype=np.uint64)
    check(lib.dp_sumcheck_prove_sharded(dev.h, group.h if group is not None else None, nv, tabs, nt, deg.ctypes.data_as(_lib.i32p), tt.ctypes.data_as(_lib.i32p),
                                        co.ctypes.data_as(_lib.u64p), len(terms), transcript.h, C.byref(pw), C.byref(pn), finals.ctypes.data_as(_lib.u64p)))
    return _take(pw, pn.value), finals


def prove_sharded_local(devs, nv, tables_per_rank, terms, transcripts):
    """dp_sumcheck_prove_sharded_local: `len(devs)` contexts of ONE process (threads + an in-memory exchange)"""
    from .api import _take
    lib = _lib.load()
    world, nt = len(devs), len(tables_per_rank[0])
    ctxs = (vp * world)(*[d.h for d in devs])
    tabs = (vp * (world * nt))(*[t.h for row in tables_per_rank for t in row])
    ts = (vp * world)(*[t.h for t in transcripts])
    deg, tt, co = _pack_terms(terms)
    pw, pn = _lib.u64p(), C.c_size_t()
    finals = np.zeros(2 * nt, dtype=np.uint64)
    check(lib.dp_sumcheck_prove_sharded_local(ctxs, world, nv, tabs, nt, deg.ctypes.data_as(_lib.i32p), tt.ctypes.data_as(_lib.i32p), co.ctypes.data_as(_lib.u64p),
                                              len(terms), ts, C.byref(pw), C.byref(pn), finals.ctypes.data_as(_lib.u64p)))
    return _take(pw, pn.value), finals
