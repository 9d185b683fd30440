"""A SECOND, independent restatement of layer 0 (test infrastructure only) — Goldilocks, its quadratic extension, the
Poseidon parameter generator (Grain LFSR), the Poseidon2 width-8 permutation as the reference wires it, the duplex sponge,
compress / hash_or_noop and BasicTranscript — in plain Python big-int arithmetic, written from the specifications and NOT
from oracle/*.hpp or deep-prove_amd/csrc (no shared code, no shared tables, different formulations throughout):

 * field: `%` on Python ints (the oracle uses u128 `%`, the product a 2^64 = 2^32 - 1 fold);
 * round constants and the internal diagonal: regenerated bit by bit from the Grain LFSR (Poseidon paper, appendix F / the
   reference generate_parameters_grain.sage: 80-bit state, b_{i+80} = b_{i+62} ^ b_{i+51} ^ b_{i+38} ^ b_{i+23} ^ b_{i+13} ^ b_i,
   160 warm-up bits, self-shrinking output, rejection sampling), with the layout of SURVEY.md Appendix B;
 * linear layers: as explicit 8x8 matrices multiplied out (the oracle and the product use the add-chain of p3's MDSMat4):
   M_E = circ(2 M4, M4) with M4 = [[2,3,1,1],[1,2,3,1],[1,1,2,3],[3,1,1,2]] (ff_ext/src/lib.rs:199-207 passes p3's MDSMat4 to
   Poseidon2ExternalMatrixGeneral), M_I = J + diag(d - 1 ... ) i.e. y_i = d_i x_i + sum(x) with the MATRIX_DIAG_8 minus one;
 * sponge: DuplexChallenger<F, P, 8, 4> semantics: overwrite-mode absorb, rate 4, squeeze pops from the back.

What this buys: parity stays "unpinned" (no reference binary or KAT exists here), but a misreading would now have to be
made twice, in two different formulations, to survive tests/test_l0_independent.py."""

P = (1 << 64) - (1 << 32) + 1
W = 7  # X^2 = 7


# ---------------------------------------------------------------- field / extension
def inv(a):
    return pow(a % P, P - 2, P)


def ext_mul(a, b):
    return ((a[0] * b[0] + W * a[1] * b[1]) % P, (a[0] * b[1] + a[1] * b[0]) % P)


def ext_add(a, b):
    return ((a[0] + b[0]) % P, (a[1] + b[1]) % P)


def ext_sub(a, b):
    return ((a[0] - b[0]) % P, (a[1] - b[1]) % P)


def ext_inv(a):
    n = inv((a[0] * a[0] - W * a[1] * a[1]) % P)  # norm
    return (a[0] * n % P, (-a[1]) * n % P)


# ---------------------------------------------------------------- Grain LFSR
def grain_field_elements(count, field=1, sbox=0, n=64, t=8, rf=8, rp=22):
    bits = []
    for val, width in ((field, 2), (sbox, 4), (n, 12), (t, 12), (rf, 10), (rp, 10)):
        bits += [(val >> (width - 1 - i)) & 1 for i in range(width)]
    bits += [1] * 30
    assert len(bits) == 80
    state = list(bits)

    def step():
        b = state[62] ^ state[51] ^ state[38] ^ state[23] ^ state[13] ^ state[0]
        state.pop(0)
        state.append(b)
        return b

    for _ in range(160):
        step()

    def shrunk_bit():
        while True:
            a, b = step(), step()
            if a:
                return b

    out = []
    while len(out) < count:
        v = 0
        for _ in range(n):
            v = (v << 1) | shrunk_bit()
        if v < P:
            out.append(v)
    return out


_G = grain_field_elements(86 + 32)
RC_EXT_INITIAL = [_G[8 * r:8 * r + 8] for r in range(4)]
RC_INTERNAL = _G[32:54]
RC_EXT_TERMINAL = [_G[54 + 8 * r:54 + 8 * r + 8] for r in range(4)]
DIAG_M1 = [(v - 1) % P for v in _G[86 + 24:86 + 32]]

# ---------------------------------------------------------------- linear layers as matrices
M4 = [[2, 3, 1, 1], [1, 2, 3, 1], [1, 1, 2, 3], [3, 1, 1, 2]]
M_E = [[(2 if (i // 4) == (j // 4) else 1) * M4[i % 4][j % 4] for j in range(8)] for i in range(8)]


def mat_vec(m, x):
    return [sum(m[i][j] * x[j] for j in range(8)) % P for i in range(8)]


def permute(state):
    """Poseidon2 (eprint 2023/323, section 4 / figure 1): M_E first, R_F/2 full rounds, R_P partial rounds, R_F/2 full rounds"""
    x = mat_vec(M_E, [v % P for v in state])
    for r in range(4):
        x = mat_vec(M_E, [pow((x[i] + RC_EXT_INITIAL[r][i]) % P, 7, P) for i in range(8)])
    for r in range(22):
        x[0] = pow((x[0] + RC_INTERNAL[r]) % P, 7, P)
        s = sum(x) % P
        x = [(DIAG_M1[i] * x[i] + s) % P for i in range(8)]
    for r in range(4):
        x = mat_vec(M_E, [pow((x[i] + RC_EXT_TERMINAL[r][i]) % P, 7, P) for i in range(8)])
    return x


# ---------------------------------------------------------------- duplex sponge, hashing, transcript
class Duplex:
    """DuplexChallenger<F, Perm, WIDTH = 8, RATE = 4>"""

    def __init__(self):
        self.state = [0] * 8
        self.inp = []
        self.out = []

    def _duplex(self):
        for i, v in enumerate(self.inp):
            self.state[i] = v
        self.inp = []
        self.state = permute(self.state)
        self.out = list(self.state[:4])

    def observe(self, v):
        self.out = []
        self.inp.append(v % P)
        if len(self.inp) == 4:
            self._duplex()

    def sample(self):
        if self.inp or not self.out:
            self._duplex()
        return self.out.pop()


def compress(x, y):
    d = Duplex()
    for v in list(x) + list(y):
        d.observe(v)
    return [d.sample() for _ in range(4)]


def hash_or_noop(elems):
    if len(elems) <= 4:
        return list(elems) + [0] * (4 - len(elems))
    d = Duplex()
    for v in elems:
        d.observe(v)
    return [d.sample() for _ in range(4)]


def bytes_to_field_elements(b):
    return [int.from_bytes(b[i:i + 8].ljust(8, b"\0"), "little") for i in range(0, len(b), 8)]


class Transcript:
    """transcript::BasicTranscript (transcript/src/basic.rs:8-54, lib.rs:42-78)"""

    def __init__(self, label=b"m2vec"):
        self.d = Duplex()
        self.append_message(label)

    def append_message(self, msg):
        for v in bytes_to_field_elements(msg):
            self.d.observe(v)

    def append_field_elements(self, vs):
        for v in vs:
            self.d.observe(v)

    def read_challenge(self):
        return (self.d.sample(), self.d.sample())

    def get_and_append_challenge(self, label):
        self.append_message(label)
        return self.read_challenge()


# ---------------------------------------------------------------- MLE helpers over the extension (little-endian index)
def eq_table(point):
    """eq(x, r) for all x in {0,1}^k, index bit t <-> point[t] (virtual_poly.rs:370-387 computes it by doubling; this is the
    per-index product)"""
    k = len(point)
    out = []
    for idx in range(1 << k):
        v = (1, 0)
        for t in range(k):
            r = point[t]
            v = ext_mul(v, r if (idx >> t) & 1 else ext_sub((1, 0), r))
        out.append(v)
    return out


def mle_eval(values, point):
    """sum_x f(x) eq(x, point) — the definition, not the fold"""
    acc = (0, 0)
    for v, e in zip(values, eq_table(point)):
        acc = ext_add(acc, ext_mul(v, e))
    return acc
