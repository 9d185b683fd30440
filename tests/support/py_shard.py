"""TEST INFRASTRUCTURE: a pure-Python shard backend (small tables only) with the interface of
deep_prove_amd.sharded.HipShard, so that the sharded-sumcheck protocol (share exchange, transcript, stage-2 merge) can be
exercised on the GPU-less CI box, including over a world_size-2 gloo process group."""
import numpy as np

from deep_prove_amd.sharded import P, e_add, e_mul, e_sub


class PyShard:
    def __init__(self, tables, terms):
        """tables: list of lists of (c0, c1) tuples (2^nv_local entries); terms: [(coeff, [table indices])]"""
        self.tabs = [list(t) for t in tables]
        self.terms = terms

    def _fold(self, r):
        self.tabs = [[e_add(t[2 * i], e_mul(e_sub(t[2 * i + 1], t[2 * i]), r)) for i in range(len(t) // 2)] for t in self.tabs]

    def round(self, r_prev):
        if r_prev is not None:
            self._fold(r_prev)
        out = []
        for _, ix in self.terms:
            k = len(ix)
            n = len(self.tabs[ix[0]])
            for tpt in range(k + 1):
                acc = (0, 0)
                for b in range(0, n, 2):
                    prod = (1, 0)
                    for j in ix:
                        a, bb = self.tabs[j][b], self.tabs[j][b + 1]
                        prod = e_mul(prod, e_add(a, e_mul(e_sub(bb, a), (tpt, 0))))
                    acc = e_add(acc, prod)
                out += [acc[0], acc[1]]
        return np.array(out, dtype=np.uint64)

    def finish(self, r_last):
        self._fold(r_last)
        return np.array([w for t in self.tabs for w in t[0]], dtype=np.uint64)

    def close(self):
        pass


def words_to_exts(words):
    return [(int(words[2 * i]), int(words[2 * i + 1])) for i in range(len(words) // 2)]
