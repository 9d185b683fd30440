"""Synthetic, already-quantised models for BASELINE.json's configs (the ONNX / float front-end of the reference is out
of scope). Tensors are SplitMix64 streams seeded per tensor: seed = 0xD33B0000 ^ (config << 32) ^ index,
values uniform in [-127, 127] (quantization MIN..MAX, zkml/src/quantization/mod.rs:28-29)."""
import math

import numpy as np

L_DENSE, L_REQUANT, L_RELU, L_CONV, L_MAXPOOL, L_FLATTEN, L_MATMUL, L_ADD, L_EMBED, L_POSITIONAL = 0, 1, 2, 3, 4, 5, 6, 7, 8, 9
L_LAYERNORM, L_SOFTMAX, L_MHA, L_GELU = 14, 15, 16, 17
L_MATMUL2, L_ADD2, L_CONCAT_MATMUL, L_QKV = 10, 11, 12, 13  # nodes of a model GRAPH: two-input MatMul / Add, ConcatMatMul, QKV
BIT_LEN = 8
FIXED_POINT_SCALE = 25  # zkml/src/layers/requant.rs:47


def splitmix64(seed, n):
    """n outputs of SplitMix64 started at `seed` (vectorised)"""
    with np.errstate(over="ignore"):
        idx = np.arange(1, n + 1, dtype=np.uint64)
        z = np.uint64(seed) + idx * np.uint64(0x9E3779B97F4A7C15)
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        return z ^ (z >> np.uint64(31))


def quantised_tensor(config, index, n):
    r = splitmix64(0xD33B0000 ^ (config << 32) ^ index, n)
    return (r % np.uint64(255)).astype(np.int64) - 127


def requant_from_multiplier(multiplier, intermediate_bit_size):
    """Requant::from_multiplier (zkml/src/layers/requant.rs:409-437), f32 arithmetic like the reference"""
    m = np.float32(multiplier)
    log_m = np.log2(m, dtype=np.float32)
    int_part = int(abs(np.trunc(log_m)))
    float_part = np.float32(log_m - np.trunc(log_m))
    epsilon = np.float32(2.0) ** float_part
    next_multiple = -(-(int_part + FIXED_POINT_SCALE) // BIT_LEN) * BIT_LEN
    fp_scale = next_multiple - int_part
    fpm = int(np.round(np.float32(epsilon) * np.float32(float(1 << fp_scale))))
    assert intermediate_bit_size + fp_scale <= 63
    return dict(right_shift=int_part, fp_scale=fp_scale, fixed_point_multiplier=fpm,
                intermediate_bit_size=intermediate_bit_size)


def inv_sqrt_table_output(eps_bits, range_check_bits, j):
    """InverseSQRTTableData::table_output (zkml/src/lookup/context.rs:147-157) for an array of table inputs: f32 arithmetic as there, `round`
    away from zero, a NaN (negative argument of the square root) becomes 0"""
    eps = np.array([eps_bits], dtype=np.uint32).view(np.float32)[0]
    shifted = np.asarray(j, dtype=np.int64) * (1 << range_check_bits)
    with np.errstate(invalid="ignore", divide="ignore"):
        x = (np.float32(1.0) / np.sqrt(shifted.astype(np.float32) / np.float32(1 << 24) + eps)) * np.float32(1 << 10)
    a = np.abs(x)
    fl = np.floor(a)
    r = np.where(a - fl >= np.float32(0.5), fl + 1, fl) * np.sign(x)
    return np.where(np.isnan(r), 0, r).astype(np.int64)


def _libm():
    """expf / logf of the C library: the quantised Softmax computes its table and its row shifts in f32 with the platform's libm (Rust's
    f32::exp / ln call the same functions), so the numpy restatement must not use numpy's own vectorised exp / log"""
    import ctypes
    import ctypes.util
    if not hasattr(_libm, "lib"):
        lib = ctypes.CDLL(ctypes.util.find_library("m") or "libm.so.6")
        for f in (lib.expf, lib.logf, lib.tanhf):
            f.restype, f.argtypes = ctypes.c_float, [ctypes.c_float]
        _libm.lib = lib
    return _libm.lib


def _round_away(x):
    """f32::round: half away from zero"""
    x = np.float32(x)
    a = np.floor(np.abs(x))
    return int((a + 1 if np.abs(x) - a >= np.float32(0.5) else a) * (1 if x >= 0 else -1))


def gelu_multiplier(in_scale):
    """GELU::quantize (zkml/src/layers/activation.rs:629-659): the integer the quantised input is multiplied by so that it counts units of 2^-12"""
    m = _round_away(np.float32(1 << 12) * np.float32(in_scale))
    assert 1 <= m <= 1 << 12  # the table has 2^(8 + ceil_log2(m)) rows, at most 2^20 (:643-648)
    return m


def gelu_table_output(scaled):
    """GELUQuantData::table_output / gelu_float (activation.rs:582-588, 623-627) for ONE scaled input: f32 after every operation, tanhf of the C
    library (Rust's f32::tanh calls the same function), round half away from zero"""
    f = np.float32
    x = f(int(scaled)) / f(1 << 12)
    cubed = f(f(x * x) * x)
    inner = f(f(np.sqrt(f(2.0) / f(np.pi))) * f(x + f(f(0.044715) * cubed)))
    g = f(f(f(0.5) * x) * f(f(1.0) + f(_libm().tanhf(inner))))
    return _round_away(f(g * f(127)))


def gelu_apply(l, x):
    """Activation::Gelu on Elements (GELU::apply, activation.rs:661-671): the table row of input * multiplier (the table ends one before its max)"""
    m = l["multiplier"]
    edge = 1 << (7 + (m - 1).bit_length())
    scaled = np.asarray(x, dtype=np.int64) * m
    assert scaled.min() >= -edge and scaled.max() < edge, "gelu: input out of range"
    lut = {}
    return np.array([lut.setdefault(int(v), gelu_table_output(int(v))) for v in scaled], dtype=np.int64)


def softmax_params(in_scale, max_context, temperature=1.0, max_abs=127):
    """Softmax::quantise (zkml/src/layers/transformer/softmax.rs:153-233) with calc_softmax_error (:323-345), f32 arithmetic: the multiplier
    to the scale 2^24, bkm (inputs beyond it are mapped to zero), the size of the exponential table, the zero tables for the remaining
    high bits, the allowable error of a row sum"""
    f = np.float32
    sf, osf, inv_temp, in_scale, ctx = f(1 << 24), f(1 << 12), f(1.0) / f(temperature), f(in_scale), f(max_context)
    scalar = _round_away(sf * in_scale)
    max_shift = _round_away(-sf * (inv_temp * np.log(ctx) + f(max_abs) * in_scale))  # input_scaling.max() = scale * domain (127 after a Requant)
    sig_min = (-max_abs * scalar + max_shift) >> 16
    min_bits = (abs(sig_min) - 1).bit_length()
    bkm_f = sf * inv_temp * (np.log(f(2) * ctx) + np.log(osf)) / f(2)
    cd = sf * inv_temp
    c = (np.exp(f(1 << 16) / cd) + np.exp(bkm_f / cd) / (f(2) * osf)) - f(1)
    err = abs(c * np.exp(f(1) / (f(2) * sf * inv_temp)) + (ctx - f(1)) * np.exp(-bkm_f / sf * inv_temp))
    bkm = _round_away(bkm_f)
    table_size = ((bkm >> 16) - 1).bit_length()
    zero_chunks = zero_vars = 0
    if min_bits > table_size:
        rem = min_bits - table_size
        zero_chunks = (rem - 1) // table_size + 1
        zero_vars = rem if zero_chunks == 1 else table_size
        zero_vars = max(zero_vars, 2)  # (this prover's lookup argument wants tables of at least four entries: one spare bit, always zero)
    return dict(scalar=scalar, temp_bits=int(np.array([inv_temp], dtype=np.float32).view(np.uint32)[0]), in_scale_bits=int(np.array([in_scale], dtype=np.float32).view(np.uint32)[0]),
                table_size=table_size, bkm=bkm, zero_chunks=zero_chunks, zero_vars=zero_vars, allowable_error=_round_away(f(err) * osf))


def softmax_table_output(l, j):
    """SoftmaxTableData::table_output (zkml/src/lookup/context.rs:111-122)"""
    prod = (1 << 16) * int(j)
    if prod >= l["bkm"]:
        return 0
    temp = np.array([l["temp_bits"]], dtype=np.uint32).view(np.float32)[0]
    return _round_away(np.float32(_libm().expf(np.float32(-prod) / (np.float32(1 << 24) * temp))) * np.float32(1 << 12))


def softmax_apply(l, x, trace=None):
    """Softmax::evaluate on quantised values (softmax.rs:455-566; calculate_shift_data :250-320; the causal AttentionMask :1590-1750) -> output;
    `trace` (a dict) receives the columns the prover commits: low, high, exp_in, exp_out, shift, zero_in[z], zero_out[z]"""
    m = _libm()
    C, R, K = l["shape"]
    x = np.asarray(x, dtype=np.int64).reshape(C * R, K)
    inv_temp, in_scale = (np.array([l[k]], dtype=np.uint32).view(np.float32)[0] for k in ("temp_bits", "in_scale_bits"))
    neg_inf = -(((l["bkm"] >> 16) + 1) << 16)
    tmask, zmask = (1 << l["table_size"]) - 1, (1 << l["zero_vars"]) - 1
    lut = {}
    out = np.zeros_like(x)
    tcols = dict(low=[], high=[], exp_in=[], exp_out=[], shift=[], zero_in=[[] for _ in range(l["zero_chunks"])], zero_out=[[] for _ in range(l["zero_chunks"])])
    for i in range(C * R):
        take = i % R + 1
        row = [int(v) for v in x[i]]
        if i % R == 0:
            shift = -row[0] * l["scalar"]
        else:
            mx = max(row[:take])
            total = np.float32(0)
            for v in row[:take]:
                total = np.float32(total + np.float32(m.expf(np.float32(np.float32(v - mx) * in_scale) / inv_temp)))
            shift = -_round_away(np.float32(1 << 24) * inv_temp * np.float32(m.logf(total))) - mx * l["scalar"]
        tcols["shift"].append(shift)
        for j in range(K):
            full = abs(row[j] * l["scalar"] + shift if j < take else neg_inf)
            tcols["low"].append(full & 255)
            tcols["high"].append((full >> 8) & 255)
            r = full >> 16
            key = r & tmask
            if key not in lut:
                lut[key] = softmax_table_output(l, key)
            o, r = lut[key], r >> l["table_size"]
            tcols["exp_in"].append(key)
            tcols["exp_out"].append(o)
            for z in range(l["zero_chunks"]):
                tcols["zero_in"][z].append(r & zmask)
                tcols["zero_out"][z].append(1 if (r & zmask) == 0 else 0)
                o, r = (o if (r & zmask) == 0 else 0), r >> l["zero_vars"]
            out[i, j] = o
    if trace is not None:
        trace.update(tcols)
    return out.reshape(-1)


def layernorm_apply(l, x):
    """LayerNorm::evaluate on quantised values (layers/transformer/layernorm.rs:394-470) -> output, (lookup input, lookup output, range-checked part)"""
    d = l["gamma"].size
    rows = np.asarray(x, dtype=np.int64).reshape(-1, d)
    s, sq = rows.sum(axis=1), (rows * rows).sum(axis=1)
    full = l["dim_size"] * l["multiplier"] * sq - l["multiplier"] * s * s
    lin = full >> l["range_check_bits"]
    inv = inv_sqrt_table_output(l["eps_bits"], l["range_check_bits"], lin)
    out = l["gamma"][None, :] * (l["dim_size"] * rows - s[:, None]) * inv[:, None] + l["beta"][None, :]
    return out.reshape(-1), (lin, inv, full & ((1 << l["range_check_bits"]) - 1))


def dense_output_bitsize(ncols):
    """Dense::output_bitsize (zkml/src/layers/dense.rs:413-419)"""
    return 2 * (BIT_LEN - 1) + max(0, (ncols - 1).bit_length()) + 1


def next_pow2(x):
    return 1 << max(0, (x - 1).bit_length())


def conv_output_bitsize(kc, kh, kw_):
    """Convolution::output_bitsize (zkml/src/layers/convolution.rs:349-353), on the unpadded filter"""
    return 2 * (BIT_LEN - 1) + max(0, (kh * kw_ * kc).bit_length())  # ceil_log2(x + 1) == bit_length(x)


class ModelBuilder:
    """Builds the already padded, already quantised model the prover consumes, doing what the reference's padding pass
    does to each node (zkml/src/padding.rs): `input_shape` is an int (vector model) or a (c, h, w) tuple (CNN)."""

    def __init__(self, input_shape, config=0):
        if isinstance(input_shape, int):
            input_shape = (input_shape,)
        self.shape_og = tuple(input_shape)                       # ShapeData.input_shape_og
        self.shape_pad = tuple(next_pow2(d) for d in input_shape)  # ShapeData.input_shape_padded
        self.input_shape_og, self.input_shape_pad = self.shape_og, self.shape_pad
        self.input_len = int(np.prod(self.shape_pad))
        self.layers = []
        self.config = config
        self._tensor_index = 0
        self._cur = self.input_len
        self._garbage = None  # GarbagePad::Convolution((og, padded)) set by flatten (padding.rs:195-203)

    def _tensor(self, n):
        t = quantised_tensor(self.config, self._tensor_index, n)
        self._tensor_index += 1
        return t

    def dense(self, out_features, in_features=None, requant=True, float_abs_max=None):
        """Dense (padded to powers of two; padding rows/cols are zero like Tensor::pad_next_power_of_two) + Requant"""
        if self._garbage is not None:
            # pad_dense + Tensor::pad_matrix_to_ignore_garbage (padding.rs:288-346, tensor.rs:1627-1675): the columns of the
            # matrix follow the PADDED (c, h, w) layout of the flattened convolution output; padding positions get zeros
            og, pad = self._garbage
            in_features = int(np.prod(og))
            r, c = next_pow2(out_features), int(np.prod(pad))
            assert c == self._cur
            m = self._tensor(out_features * in_features).reshape(out_features, *og)
            w4 = np.zeros((r,) + tuple(pad), dtype=np.int64)
            w4[:out_features, :og[0], :og[1], :og[2]] = m
            w = w4.reshape(r, c)
            self._garbage = None
        else:
            in_features = in_features or self.shape_og[0]
            r, c = next_pow2(out_features), next_pow2(in_features)
            assert c == self._cur
            w = np.zeros((r, c), dtype=np.int64)
            w[:out_features, :in_features] = self._tensor(out_features * in_features).reshape(out_features, in_features)
        self.shape_og, self.shape_pad = (out_features,), (r,)
        b = np.zeros(r, dtype=np.int64)
        b[:out_features] = self._tensor(out_features)
        self.layers.append(dict(kind=L_DENSE, nrows=r, ncols=c, weights=w, bias=b))
        self._cur = r
        if requant:
            # AbsoluteMax strategy with default input/output scaling 2/254: m = S_w = max|w_float| / 127. |w_float| is
            # taken as g/sqrt(fan_in) (g = 1: PyTorch's default Linear init bound; g = 2.5 on the wide layers keeps the
            # synthetic activations from collapsing to zero, so every lookup table sees a spread of values)
            gain = 1.0 if in_features <= 4 else 2.5
            amax = float_abs_max if float_abs_max is not None else gain / math.sqrt(in_features)
            rq = requant_from_multiplier(amax / 127.0, dense_output_bitsize(c))
            self.layers.append(dict(kind=L_REQUANT, **rq))
        return self

    def matmul(self, out_features, bias=True, requant=True, transpose_b=False):
        """MatMul::new_constant(right, bias) (layers/matrix_mul.rs:176): the activation is a [seq][features] matrix (both padded to
        powers of two), the constant RIGHT matrix [features][out_features] — a Linear layer applied to every row of a sequence, the
        building block of the transformer layers — followed by its Requant node"""
        assert len(self.shape_og) == 2, "matmul needs a [seq, features] activation"
        s_og, k_og = self.shape_og
        s, k = self.shape_pad
        n = next_pow2(out_features)
        assert s * k == self._cur and s >= 2
        w = np.zeros((k, n), dtype=np.int64)
        w[:k_og, :out_features] = self._tensor(k_og * out_features).reshape(k_og, out_features)
        b = None
        if bias:
            b = np.zeros(n, dtype=np.int64)
            b[:out_features] = self._tensor(out_features)
        if transpose_b:  # Config::TransposeB: the constant matrix is given as [out_features][features] (e.g. tied embeddings) and used transposed
            w = np.ascontiguousarray(w.T)
        self.layers.append(dict(kind=L_MATMUL, nrows=k, ncols=n, weights=w, bias=b, transpose_b=transpose_b))
        self.shape_og, self.shape_pad = (s_og, out_features), (s, n)
        self._cur = s * n
        if requant:
            gain = 1.0 if k_og <= 4 else 2.5
            rq = requant_from_multiplier(gain / math.sqrt(k_og) / 127.0, dense_output_bitsize(k))
            self.layers.append(dict(kind=L_REQUANT, **rq))
        return self

    def layernorm(self, eps=1e-5, requant=True):
        """LayerNorm over the last dimension of a [seq][features] activation, quantised as LayerNorm::quantise does it (layers/transformer/
        layernorm.rs:140-257: the multiplier of the inverse-square-root input, the bits that are shifted away and range checked, the rescaled
        epsilon), followed by the shift-only Requant of Requant::new_shift (:473-513). gamma / beta are zero on the padding of the dimension."""
        assert len(self.shape_og) == 2, "layernorm needs a [seq, features] activation"
        (s_og, n_og), (s, d) = self.shape_og, self.shape_pad
        assert s >= 4 and d >= 2 and next_pow2(n_og) == d
        in_scale = np.float32(1.0 / 127.0)
        multiplier = int(np.round(np.float32(1 << 24) * in_scale * in_scale))
        clog = lambda v: max(0, (int(v) - 1).bit_length())
        rcb = 2 * (clog(n_og) + BIT_LEN - 1) + clog(multiplier) + 1 - 2 * (BIT_LEN - 1)
        eps_bits = int(np.array([np.float32(n_og * n_og) * np.float32(eps)], dtype=np.float32).view(np.uint32)[0])
        gamma, beta = np.zeros(d, dtype=np.int64), np.zeros(d, dtype=np.int64)
        gamma[:n_og] = self._tensor(n_og)
        beta[:n_og] = self._tensor(n_og) * 4096
        self.layers.append(dict(kind=L_LAYERNORM, dim=d, dim_size=n_og, multiplier=multiplier, eps_bits=eps_bits, range_check_bits=rcb,
                                top_chunk_scalar_log=(BIT_LEN - rcb % BIT_LEN) % BIT_LEN, gamma=gamma, beta=beta))
        if requant:
            max_lut = int(np.abs(inv_sqrt_table_output(eps_bits, rcb, np.arange(-(1 << 14), 1 << 14))).max())
            ibs = max(2 * (BIT_LEN - 1) + clog(n_og) + 1 + clog(max_lut), clog(int(np.abs(beta).max()) or 1)) + 1
            right_shift = ibs - 16  # (the scale of the next layer is chosen so that the clamping table has 2^16 entries)
            fp_scale = -right_shift % BIT_LEN
            self.layers.append(dict(kind=L_REQUANT, right_shift=right_shift, fp_scale=fp_scale, fixed_point_multiplier=1 << fp_scale, intermediate_bit_size=ibs))
        return self

    def softmax(self, in_scale=1.0 / 127.0, temperature=1.0):
        """Softmax over the last dimension of [heads][n][n] attention scores under the causal mask (layers/transformer/softmax.rs), quantised as
        Softmax::quantise does; the output has the scale 2^-12"""
        assert len(self.shape_og) == 3 and self.shape_pad[1] == self.shape_pad[2], "softmax needs a [heads, n, n] activation"
        self.layers.append(dict(kind=L_SOFTMAX, shape=self.shape_pad, **softmax_params(in_scale, self.shape_pad[2], temperature)))
        return self

    def embeddings(self, vocab, emb):
        """Embeddings (layers/transformer/embeddings.rs): the first layer of a model whose input is a vector of token ids; the table is
        [vocab][emb], both padded to powers of two; the output is the [tokens][emb] matrix of the looked-up rows (no requant: the
        table holds quantised values already)"""
        assert not self.layers and len(self.shape_og) == 1, "embeddings must be the first layer, over a 1-d token vector"
        v, e = next_pow2(vocab), next_pow2(emb)
        t = np.zeros((v, e), dtype=np.int64)
        t[:vocab, :emb] = self._tensor(vocab * emb).reshape(vocab, emb)
        self.layers.append(dict(kind=L_EMBED, nrows=v, ncols=e, table=t, vocab=vocab))
        self.shape_og, self.shape_pad = (self.shape_og[0], emb), (self.shape_pad[0], e)
        self._cur = self.shape_pad[0] * e
        return self

    def positional(self, max_positions, left=1, right=1, requant=True):
        """Positional::Learned (layers/transformer/positional.rs): a [max_positions][features] table is committed, its first `tokens` rows
        are added to the [tokens][features] activation (out = left * x + right * table[:tokens]); max_positions >= tokens (padded)"""
        assert len(self.shape_og) == 2
        s_og, k_og = self.shape_og
        s, k = self.shape_pad
        mp = next_pow2(max_positions)
        assert mp >= s
        t = np.zeros((mp, k), dtype=np.int64)
        t[:max_positions, :k_og] = self._tensor(max_positions * k_og).reshape(max_positions, k_og)
        self.layers.append(dict(kind=L_POSITIONAL, left=int(left), right=int(right), nrows=mp, ncols=k, table=t))
        if requant:
            bits = 8 + int(math.ceil(math.log2(left + right)))
            self.layers.append(dict(kind=L_REQUANT, **requant_from_multiplier(1.0 / (left + right), bits)))
        return self

    def add_const(self, left=1, right=1, requant=True):
        """Add::new_with(operand) (layers/add.rs:72-78): out = left * x + right * operand with a constant operand as long as the
        activation (learned positional embeddings are added this way, transformer/positional.rs); zeros at the padding positions.
        Followed by a Requant that brings the sum back into the 8-bit range"""
        og, pad = self.shape_og, self.shape_pad
        c = np.zeros(pad, dtype=np.int64)
        c[tuple(slice(0, d) for d in og)] = self._tensor(int(np.prod(og))).reshape(og)
        self.layers.append(dict(kind=L_ADD, left=int(left), right=int(right), operand=c.reshape(-1)))
        if requant:
            # |left x + right c| <= 127 (left + right): one more bit per doubling; multiplier 1 / (left + right)
            bits = 8 + int(math.ceil(math.log2(left + right)))
            self.layers.append(dict(kind=L_REQUANT, **requant_from_multiplier(1.0 / (left + right), bits)))
        return self

    def relu(self):
        self.layers.append(dict(kind=L_RELU))
        return self

    def gelu(self, in_scale=1.0 / 128.0):
        """Activation::Gelu behind a Requant whose output carries `in_scale` (layers/activation.rs:559-671)"""
        self.layers.append(dict(kind=L_GELU, multiplier=gelu_multiplier(in_scale)))
        return self

    def conv(self, out_channels, kernel, requant=True):
        """Convolution (stride 1, no padding) as pad_conv + into_padded_and_ffted lay it out (padding.rs:218-260,
        convolution.rs:290-301, tensor.rs:409-431): filter zero padded to powers of two in every dimension, nw = the
        padded input side; followed by its Requant node"""
        assert len(self.shape_og) == 3
        kc, h, w_ = self.shape_og
        assert h == w_ and self.shape_pad[1] == self.shape_pad[2]
        kw, kx, rnw = next_pow2(out_channels), next_pow2(kc), next_pow2(kernel)
        assert kx == self.shape_pad[0]
        nw = next_pow2(self.shape_pad[1] - rnw + 1)
        assert nw == self.shape_pad[1], "FFT convolution needs padded kernel <= half of the padded input side"
        f = np.zeros((kw, kx, rnw, rnw), dtype=np.int64)
        f[:out_channels, :kc, :kernel, :kernel] = self._tensor(out_channels * kc * kernel * kernel).reshape(out_channels, kc, kernel, kernel)
        b = np.zeros(kw, dtype=np.int64)
        b[:out_channels] = self._tensor(out_channels)
        unp_out = (out_channels, h - kernel + 1, w_ - kernel + 1)
        self.layers.append(dict(kind=L_CONV, kw=kw, kx=kx, real_nw=rnw, nw=nw, unp_out=unp_out, kernel=kernel, filter=f, bias=b))
        self.shape_og, self.shape_pad = unp_out, (kw, nw, nw)
        self._cur = kw * nw * nw
        if requant:
            fan_in = kc * kernel * kernel
            rq = requant_from_multiplier((1.0 / math.sqrt(fan_in)) / 127.0, conv_output_bitsize(kc, kernel, kernel))
            self.layers.append(dict(kind=L_REQUANT, **rq))
        return self

    def maxpool(self):
        """Maxpool2D kernel 2 stride 2 (layers/pooling.rs, padding.rs:205-216)"""
        assert len(self.shape_og) == 3
        self.layers.append(dict(kind=L_MAXPOOL, pin=self.shape_pad))
        self.shape_og = (self.shape_og[0], (self.shape_og[1] - 2) // 2 + 1, (self.shape_og[2] - 2) // 2 + 1)
        self.shape_pad = (self.shape_pad[0], self.shape_pad[1] // 2, self.shape_pad[2] // 2)
        self._cur = int(np.prod(self.shape_pad))
        return self

    def flatten(self):
        self._garbage = (self.shape_og, self.shape_pad)
        self.layers.append(dict(kind=L_FLATTEN))
        return self

    def blob(self):
        out = [self.input_len, len(self.layers)]
        parts = [np.array(out, dtype=np.int64)]
        for l in self.layers:
            if l["kind"] == L_DENSE:
                parts.append(np.array([L_DENSE, l["nrows"], l["ncols"]], dtype=np.int64))
                parts.append(l["weights"].reshape(-1))
                parts.append(l["bias"])
            elif l["kind"] == L_POSITIONAL:
                parts.append(np.array([L_POSITIONAL, l["left"], l["right"], l["nrows"], l["ncols"]], dtype=np.int64))
                parts.append(l["table"].reshape(-1))
            elif l["kind"] == L_EMBED:
                parts.append(np.array([L_EMBED, l["nrows"], l["ncols"]], dtype=np.int64))
                parts.append(l["table"].reshape(-1))
            elif l["kind"] == L_ADD:
                parts.append(np.array([L_ADD, l["left"], l["right"], l["operand"].size], dtype=np.int64))
                parts.append(l["operand"])
            elif l["kind"] == L_MATMUL:
                parts.append(np.array([L_MATMUL, l["nrows"], l["ncols"], (0 if l["bias"] is None else 1) | (2 if l["transpose_b"] else 0)], dtype=np.int64))
                parts.append(l["weights"].reshape(-1))
                if l["bias"] is not None:
                    parts.append(l["bias"])
            elif l["kind"] == L_REQUANT:
                parts.append(np.array([L_REQUANT, l["right_shift"], l["fp_scale"], l["fixed_point_multiplier"],
                                       l["intermediate_bit_size"]], dtype=np.int64))
            elif l["kind"] == L_CONV:
                parts.append(np.array([L_CONV, l["kw"], l["kx"], l["real_nw"], l["nw"], *l["unp_out"]], dtype=np.int64))
                parts.append(l["filter"].reshape(-1))
                parts.append(l["bias"])
            elif l["kind"] == L_MAXPOOL:
                parts.append(np.array([L_MAXPOOL, *l["pin"]], dtype=np.int64))
            elif l["kind"] == L_SOFTMAX:
                parts.append(np.array([L_SOFTMAX, *l["shape"], l["scalar"], l["temp_bits"], l["in_scale_bits"], l["table_size"], l["bkm"], l["zero_chunks"], l["zero_vars"], l["allowable_error"]], dtype=np.int64))
            elif l["kind"] == L_LAYERNORM:
                parts.append(np.array([L_LAYERNORM, l["dim"], l["dim_size"], l["multiplier"], l["eps_bits"], l["range_check_bits"], l["top_chunk_scalar_log"]], dtype=np.int64))
                parts.append(l["gamma"])
                parts.append(l["beta"])
            elif l["kind"] == L_GELU:
                parts.append(np.array([L_GELU, l["multiplier"]], dtype=np.int64))
            else:
                parts.append(np.array([l["kind"]], dtype=np.int64))
        return np.concatenate(parts)

    def input(self, index=1000):
        """a synthetic input, already padded (zeros outside the unpadded shape, like Tensor::pad_next_power_of_two)"""
        og, pad = self.input_shape_og, self.input_shape_pad
        if self.layers and self.layers[0]["kind"] == L_EMBED:  # token ids (padding positions hold token 0, like a padded prompt)
            x = np.zeros(pad, dtype=np.int64)
            x[:og[0]] = np.abs(quantised_tensor(self.config, index, og[0])) % self.layers[0]["vocab"]
            return x
        x = np.zeros(pad, dtype=np.int64)
        x[tuple(slice(0, d) for d in og)] = quantised_tensor(self.config, index, int(np.prod(og))).reshape(og)
        return x.reshape(-1)

    def run(self, x):
        """quantised inference in numpy (Model::run semantics: Dense matvec + bias, Requant::apply, Relu::apply)"""
        cur = np.asarray(x, dtype=np.int64)
        for l in self.layers:
            if l["kind"] == L_DENSE:
                cur = l["weights"] @ cur + l["bias"]
            elif l["kind"] == L_POSITIONAL:
                cur = l["left"] * cur + l["right"] * l["table"].reshape(-1)[:cur.size]
            elif l["kind"] == L_EMBED:
                cur = l["table"][cur].reshape(-1)
            elif l["kind"] == L_ADD:
                cur = l["left"] * cur + l["right"] * l["operand"]
            elif l["kind"] == L_MATMUL:
                y = cur.reshape(-1, l["nrows"]) @ (l["weights"].T if l["transpose_b"] else l["weights"])
                cur = (y + l["bias"] if l["bias"] is not None else y).reshape(-1)
            elif l["kind"] == L_REQUANT:
                sh = l["fp_scale"] + l["right_shift"]
                cur = np.clip((cur * l["fixed_point_multiplier"] + (1 << (sh - 1))) >> sh, -127, 127)
            elif l["kind"] == L_RELU:
                cur = np.maximum(cur, 0)
            elif l["kind"] == L_GELU:
                cur = gelu_apply(l, cur)
            elif l["kind"] == L_LAYERNORM:
                cur = layernorm_apply(l, cur)[0]
            elif l["kind"] == L_SOFTMAX:
                cur = softmax_apply(l, cur)
            elif l["kind"] == L_CONV:
                # direct correlation on the padded tensors; everything outside the unpadded output shape is cleared
                kw, kx, k, nw = l["kw"], l["kx"], l["kernel"], l["nw"]
                x = cur.reshape(kx, nw, nw)
                oc, oh, ow = l["unp_out"]
                out = np.zeros((kw, nw, nw), dtype=np.int64)
                for a in range(k):
                    for b in range(k):
                        out[:, :oh, :ow] += np.einsum("oc,cyx->oyx", l["filter"][:, :, a, b], x[:, a:a + oh, b:b + ow])
                out += l["bias"][:, None, None]
                mask = np.zeros_like(out)
                mask[:oc, :oh, :ow] = 1
                cur = (out * mask).reshape(-1)
            elif l["kind"] == L_MAXPOOL:
                c, h, w_ = l["pin"]
                cur = cur.reshape(c, h // 2, 2, w_ // 2, 2).max(axis=(2, 4)).reshape(-1)
            elif l["kind"] == L_FLATTEN:
                pass
        return cur


class GraphBuilder:
    """A model as a GRAPH of nodes (zkml/src/layers/provable/mod.rs:195-565): nodes with two inputs (MatMul / Add of two tensors, ConcatMatMul),
    a node with three outputs (QKV), several input and output tensors. Tensors are already padded and quantised; an edge is (node, slot) with
    node = -1 for input tensor `slot` of the model. The reference proves graphs in which every tensor is read exactly once."""

    def __init__(self, input_lens, config=0):
        self.input_lens = [int(n) for n in input_lens]
        self.input_len = sum(self.input_lens)
        self.nodes = []       # (dict, [edges])
        self.outputs = None   # default: output 0 of the last node
        self.config = config
        self._tensor_index = 0

    def _tensor(self, n):
        t = quantised_tensor(self.config, self._tensor_index, n)
        self._tensor_index += 1
        return t

    def _add(self, layer, edges):
        self.nodes.append((layer, [tuple(e) for e in edges]))
        return len(self.nodes) - 1

    def qkv(self, src, k, n):
        """QKV (layers/transformer/qkv.rs): X [s][k] -> X W_q + b_q, X W_k + b_k, X W_v + b_v, outputs (id, 0..2)"""
        w = self._tensor(3 * k * n).reshape(3, k, n)
        b = self._tensor(3 * n).reshape(3, n)
        return self._add(dict(kind=L_QKV, nrows=k, ncols=n, weights=w, bias=b), [src])

    def matmul2(self, a, b, k, n, transpose_b=False):
        """MatMul of two input tensors (layers/matrix_mul.rs): [s][k] x [k][n] ([n][k] with transpose_b)"""
        return self._add(dict(kind=L_MATMUL2, nrows=k, ncols=n, transpose_b=transpose_b), [a, b])

    def add2(self, a, b, left=1, right=1):
        return self._add(dict(kind=L_ADD2, left=left, right=right), [a, b])

    def concat_matmul(self, a, b, a_shape, b_shape, left, right, perm=None):
        """ConcatMatMul (layers/concat_matmul.rs): rank-3 inputs; left / right = (concat, mat_mul, output) axis of each; perm = permutation
        of the [concat][rows][cols] result"""
        return self._add(dict(kind=L_CONCAT_MATMUL, a_shape=tuple(a_shape), b_shape=tuple(b_shape), left=tuple(left), right=tuple(right),
                              perm=None if perm is None else tuple(perm)), [a, b])

    def matmul_const(self, src, k, n, bias=True):
        w = self._tensor(k * n).reshape(k, n)
        b = self._tensor(n) if bias else None
        return self._add(dict(kind=L_MATMUL, nrows=k, ncols=n, weights=w, bias=b, transpose_b=False), [src])

    def requant(self, src, multiplier, intermediate_bit_size):
        return self._add(dict(kind=L_REQUANT, **requant_from_multiplier(multiplier, intermediate_bit_size)), [src])

    def requant_shift(self, src, right_shift, intermediate_bit_size):
        """Requant::new_shift (layers/transformer/layernorm.rs:473-513): a power-of-two multiplier"""
        fp = -right_shift % BIT_LEN
        return self._add(dict(kind=L_REQUANT, right_shift=right_shift, fp_scale=fp, fixed_point_multiplier=1 << fp, intermediate_bit_size=intermediate_bit_size), [src])

    def layernorm(self, src, dim, eps=1e-5):
        """LayerNorm over the last dimension `dim` (a power of two) of a [rows][dim] tensor, as ModelBuilder.layernorm; returns (node, the
        intermediate bit size of its output)"""
        mb = ModelBuilder((4, dim), self.config)
        mb._tensor_index = self._tensor_index
        mb.layernorm(eps=eps, requant=True)
        self._tensor_index = mb._tensor_index
        return self._add(mb.layers[0], [src]), mb.layers[1]["intermediate_bit_size"]

    def softmax(self, src, shape, in_scale, temperature=1.0):
        """Softmax over the last dimension of a [heads][n][n] tensor under the causal mask (layers/transformer/softmax.rs)"""
        assert len(shape) == 3 and shape[1] == shape[2]
        return self._add(dict(kind=L_SOFTMAX, shape=tuple(shape), **softmax_params(in_scale, shape[2], temperature)), [src])

    def mha(self, q, k, v, seq, heads, head_dim, in_scale, max_abs):
        """Mha (layers/transformer/mha.rs:147-186) as ONE node over Q, K, V ([seq][heads * head_dim] each): Q_h K_h^T per head, the Softmax with
        scale 1 / sqrt(head_dim) directly on the products (in_scale = scale of Q times scale of K, max_abs = qk.output_domain()), probabilities
        times V_h laid out as [seq][heads][head_dim]; the output carries the scale 2^-12 of the probabilities times the scale of V"""
        return self._add(dict(kind=L_MHA, shape=(seq, heads, head_dim), **softmax_params(in_scale, seq, 1.0 / math.sqrt(head_dim), max_abs)), [q, k, v])

    def relu(self, src):
        return self._add(dict(kind=L_RELU), [src])

    def gelu(self, src, in_scale=1.0 / 128.0):
        return self._add(dict(kind=L_GELU, multiplier=gelu_multiplier(in_scale)), [src])

    def set_outputs(self, edges):
        self.outputs = [tuple(e) for e in edges]
        return self

    def blob(self):
        outs = self.outputs if self.outputs is not None else [(len(self.nodes) - 1, 0)]
        head = [self.input_len, -len(self.nodes), len(self.input_lens), *self.input_lens, len(outs)]
        for e in outs:
            head += list(e)
        parts = [np.array(head, dtype=np.int64)]
        for l, edges in self.nodes:
            pre = [l["kind"], len(edges)]
            for e in edges:
                pre += list(e)
            k = l["kind"]
            if k == L_QKV:
                parts += [np.array(pre + [l["nrows"], l["ncols"]], dtype=np.int64), l["weights"].reshape(-1), l["bias"].reshape(-1)]
            elif k == L_MATMUL2:
                parts.append(np.array(pre + [l["nrows"], l["ncols"], 2 if l["transpose_b"] else 0], dtype=np.int64))
            elif k == L_ADD2:
                parts.append(np.array(pre + [l["left"], l["right"]], dtype=np.int64))
            elif k == L_CONCAT_MATMUL:
                tail = [1, *l["perm"]] if l["perm"] is not None else [0]
                parts.append(np.array(pre + [*l["a_shape"], *l["b_shape"], *l["left"], *l["right"], *tail], dtype=np.int64))
            elif k == L_MATMUL:
                parts.append(np.array(pre + [l["nrows"], l["ncols"], 0 if l["bias"] is None else 1], dtype=np.int64))
                parts.append(l["weights"].reshape(-1))
                if l["bias"] is not None:
                    parts.append(l["bias"])
            elif k == L_REQUANT:
                parts.append(np.array(pre + [l["right_shift"], l["fp_scale"], l["fixed_point_multiplier"], l["intermediate_bit_size"]], dtype=np.int64))
            elif k == L_LAYERNORM:
                parts += [np.array(pre + [l["dim"], l["dim_size"], l["multiplier"], l["eps_bits"], l["range_check_bits"], l["top_chunk_scalar_log"]], dtype=np.int64), l["gamma"], l["beta"]]
            elif k == L_SOFTMAX:
                parts.append(np.array(pre + [*l["shape"], l["scalar"], l["temp_bits"], l["in_scale_bits"], l["table_size"], l["bkm"], l["zero_chunks"], l["zero_vars"], l["allowable_error"]], dtype=np.int64))
            elif k == L_MHA:
                parts.append(np.array(pre + [*l["shape"], l["scalar"], l["temp_bits"], l["in_scale_bits"], l["table_size"], l["bkm"], l["zero_chunks"], l["zero_vars"], l["allowable_error"]], dtype=np.int64))
            elif k == L_RELU:
                parts.append(np.array(pre, dtype=np.int64))
            elif k == L_GELU:
                parts.append(np.array(pre + [l["multiplier"]], dtype=np.int64))
            else:
                raise ValueError("GraphBuilder: unsupported node kind")
        return np.concatenate(parts)

    def input(self, index=1000, amplitude=127):
        x = quantised_tensor(self.config, index, self.input_len)
        return x if amplitude == 127 else x % (2 * amplitude + 1) - amplitude

    def run(self, x):
        """quantised inference in numpy: the concatenated output tensors"""
        x = np.asarray(x, dtype=np.int64)
        offs = np.concatenate([[0], np.cumsum(self.input_lens)])
        vals = {}

        def get(e):
            return x[offs[e[1]]:offs[e[1] + 1]] if e[0] < 0 else vals[e]

        for i, (l, edges) in enumerate(self.nodes):
            a = get(edges[0])
            k = l["kind"]
            if k == L_QKV:
                xs = a.reshape(-1, l["nrows"])
                for w in range(3):
                    vals[(i, w)] = (xs @ l["weights"][w] + l["bias"][w]).reshape(-1)
                continue
            if k == L_MATMUL2:
                b = get(edges[1])
                bm = b.reshape(l["ncols"], l["nrows"]).T if l["transpose_b"] else b.reshape(l["nrows"], l["ncols"])
                y = (a.reshape(-1, l["nrows"]) @ bm).reshape(-1)
            elif k == L_ADD2:
                y = l["left"] * a + l["right"] * get(edges[1])
            elif k == L_CONCAT_MATMUL:
                b = get(edges[1])
                ta = a.reshape(l["a_shape"]).transpose(l["left"][0], l["left"][2], l["left"][1])     # [concat][rows][inner]
                tb = b.reshape(l["b_shape"]).transpose(l["right"][0], l["right"][1], l["right"][2])  # [concat][inner][cols]
                r = np.einsum("crm,cmn->crn", ta, tb)
                y = (r if l["perm"] is None else r.transpose(l["perm"])).reshape(-1)
            elif k == L_MATMUL:
                y = a.reshape(-1, l["nrows"]) @ l["weights"]
                y = (y + l["bias"] if l["bias"] is not None else y).reshape(-1)
            elif k == L_REQUANT:
                sh = l["fp_scale"] + l["right_shift"]
                y = np.clip((a * l["fixed_point_multiplier"] + (1 << (sh - 1))) >> sh, -127, 127)
            elif k == L_RELU:
                y = np.maximum(a, 0)
            elif k == L_GELU:
                y = gelu_apply(l, a)
            elif k == L_LAYERNORM:
                y = layernorm_apply(l, a)[0]
            elif k == L_SOFTMAX:
                y = softmax_apply(l, a)
            elif k == L_MHA:
                S, H, D = l["shape"]
                qh, kh, vh = (get(e).reshape(S, H, D).transpose(1, 0, 2) for e in edges)  # [h][s][d]
                scores = np.einsum("hsd,htd->hst", qh, kh)
                probs = softmax_apply(dict(l, shape=(H, S, S)), scores.reshape(-1)).reshape(H, S, S)
                y = np.einsum("hst,htd->hsd", probs, vh).transpose(1, 0, 2).reshape(-1)
            vals[(i, 0)] = y
        outs = self.outputs if self.outputs is not None else [(len(self.nodes) - 1, 0)]
        return np.concatenate([vals[e] for e in outs])


def attention_block(seq, emb, heads, head_dim, config):
    """One attention block WITHOUT softmax, as a graph: X -> QKV -> Requant (x3); scores_h = Q_h K_h^T (ConcatMatMul over [s][h][d] tensors,
    the heads as the concat axis) -> Requant; out_h = scores_h V_h laid out as [s][h][d] -> Requant; output projection (MatMul with a
    constant matrix) -> Requant; + the second input tensor (the residual stream enters as its own input: the reference proves graphs whose
    tensors have ONE reader each). layers/transformer/qkv.rs, layers/concat_matmul.rs, layers/matrix_mul.rs, layers/add.rs"""
    n = heads * head_dim
    g = GraphBuilder([seq * emb, seq * emb], config)
    q = g.qkv((-1, 0), emb, n)
    gain = 2.5 / math.sqrt(emb) / 127.0
    rq = [g.requant((q, w), gain, dense_output_bitsize(emb)) for w in range(3)]
    sc = g.concat_matmul((rq[0], 0), (rq[1], 0), (seq, heads, head_dim), (seq, heads, head_dim), (1, 2, 0), (1, 2, 0))
    sr = g.requant((sc, 0), 2.5 / math.sqrt(head_dim) / 127.0, dense_output_bitsize(head_dim))
    av = g.concat_matmul((sr, 0), (rq[2], 0), (heads, seq, seq), (seq, heads, head_dim), (0, 2, 1), (1, 0, 2), perm=(1, 0, 2))
    ar = g.requant((av, 0), 2.5 / math.sqrt(seq) / 127.0, dense_output_bitsize(seq))
    pr = g.matmul_const((ar, 0), n, emb)
    prq = g.requant((pr, 0), 2.5 / math.sqrt(n) / 127.0, dense_output_bitsize(n))
    g.add2((prq, 0), (-1, 1))
    return g


def transformer_block(seq, emb, heads, head_dim, config):
    """The attention half of a pre-LN transformer block, every layer of it proved: X -> LayerNorm -> Requant (shift) -> QKV -> Requant (x3);
    scores_h = Q_h K_h^T per head ([h][s][s], ConcatMatMul) -> Requant -> Softmax under the causal mask -> probs_h V_h laid out as [s][h][d]
    (ConcatMatMul; the probabilities carry the scale 2^-12) -> Requant -> output projection -> Requant; + the residual (a second input tensor:
    the reference proves graphs whose tensors have one reader each). layers/transformer/{layernorm,qkv,softmax}.rs, layers/concat_matmul.rs,
    layers/matrix_mul.rs, layers/add.rs — what the reference's Mha layer (transformer/mha.rs) bundles into one node, as separate nodes."""
    n = heads * head_dim
    g = GraphBuilder([seq * emb, seq * emb], config)
    ln, ibs = g.layernorm((-1, 0), emb)
    lr = g.requant_shift((ln, 0), ibs - 16, ibs)
    q = g.qkv((lr, 0), emb, n)
    rq = [g.requant((q, w), 2.5 / math.sqrt(emb) / 127.0, dense_output_bitsize(emb)) for w in range(3)]
    sc = g.concat_matmul((rq[0], 0), (rq[1], 0), (seq, heads, head_dim), (seq, heads, head_dim), (1, 2, 0), (1, 2, 0))
    sr = g.requant((sc, 0), 2.5 / math.sqrt(head_dim) / 127.0, dense_output_bitsize(head_dim))
    sm = g.softmax((sr, 0), (heads, seq, seq), in_scale=8.0 / 127.0)
    av = g.concat_matmul((sm, 0), (rq[2], 0), (heads, seq, seq), (seq, heads, head_dim), (0, 2, 1), (1, 0, 2), perm=(1, 0, 2))
    ar = g.requant((av, 0), 1.0 / 4096.0, 12 + BIT_LEN + max(0, (seq - 1).bit_length()) + 1)
    pr = g.matmul_const((ar, 0), n, emb)
    prq = g.requant((pr, 0), 2.5 / math.sqrt(n) / 127.0, dense_output_bitsize(n))
    g.add2((prq, 0), (-1, 1))
    return g


def mha_block(seq, emb, heads, head_dim, config):
    """The attention half of a transformer block with the reference's Mha layer as ONE node (transformer/mha.rs): X -> LayerNorm -> Requant ->
    QKV -> Requant (x3) -> Mha (Q K^T, Softmax with scale 1 / sqrt(head_dim) straight on the products, times V) -> Requant -> output projection ->
    Requant; + the residual (a second input tensor)"""
    n = heads * head_dim
    g = GraphBuilder([seq * emb, seq * emb], config)
    ln, ibs = g.layernorm((-1, 0), emb)
    lr = g.requant_shift((ln, 0), ibs - 16, ibs)
    q = g.qkv((lr, 0), emb, n)
    rq = [g.requant((q, w), 2.5 / math.sqrt(emb) / 127.0, dense_output_bitsize(emb)) for w in range(3)]
    s_in = 1.0 / 16.0  # scale of Q and of K: the products carry s_in^2
    att = g.mha((rq[0], 0), (rq[1], 0), (rq[2], 0), seq, heads, head_dim, s_in * s_in, (1 << dense_output_bitsize(head_dim)) - 1)
    ar = g.requant((att, 0), 1.0 / 4096.0, 12 + BIT_LEN + max(0, (seq - 1).bit_length()) + 1)
    pr = g.matmul_const((ar, 0), n, emb)
    prq = g.requant((pr, 0), 2.5 / math.sqrt(n) / 127.0, dense_output_bitsize(n))
    g.add2((prq, 0), (-1, 1))
    return g


def transformer_layer(seq, emb, heads, head_dim, ffn, config, gelu=False):
    """One whole pre-LN transformer layer as ONE graph, every node proved: the attention half of mha_block (LayerNorm -> QKV -> Mha -> projection
    -> + residual) and the feed-forward half (LayerNorm -> Linear(emb, ffn) -> ReLU -> Linear(ffn, emb) -> + residual). The reference proves graphs
    whose tensors have ONE reader each (claims_for_node, provable/mod.rs:243-248), so every second use of a tensor enters as an input tensor of
    its own: inputs = X (normalised by the first LayerNorm), X once more (the residual added to the attention output), H (what the second
    LayerNorm normalises — in a running model the hidden state X + attention, whose one reader inside the graph is the LAST Add). gelu=False: ReLU
    stands in for the GELU of the feed-forward half (what golden case 14 and the bench line prove). The reference's GELU prover files a claim its verifier
    does not check (activation.rs:405-430 against :495-505), so its proofs only verify for columns of at most 2^7 entries; gelu=True puts this library's
    GELU there (csrc/zkml.h prove_relu), input scale 1 / 128. layers/transformer/{layernorm,qkv,mha}.rs, layers/matrix_mul.rs, layers/activation.rs, layers/add.rs"""
    n = heads * head_dim
    g = GraphBuilder([seq * emb] * 3, config)
    ln, ibs = g.layernorm((-1, 0), emb)
    lr = g.requant_shift((ln, 0), ibs - 16, ibs)
    q = g.qkv((lr, 0), emb, n)
    rq = [g.requant((q, w), 2.5 / math.sqrt(emb) / 127.0, dense_output_bitsize(emb)) for w in range(3)]
    s_in = 1.0 / 16.0
    att = g.mha((rq[0], 0), (rq[1], 0), (rq[2], 0), seq, heads, head_dim, s_in * s_in, (1 << dense_output_bitsize(head_dim)) - 1)
    ar = g.requant((att, 0), 1.0 / 4096.0, 12 + BIT_LEN + max(0, (seq - 1).bit_length()) + 1)
    pr = g.matmul_const((ar, 0), n, emb)
    prq = g.requant((pr, 0), 2.5 / math.sqrt(n) / 127.0, dense_output_bitsize(n))
    res1 = g.add2((prq, 0), (-1, 1))
    # feed-forward half on the third input tensor (= the first output; a tensor has one reader)
    ln2, ibs2 = g.layernorm((-1, 2), emb)
    lr2 = g.requant_shift((ln2, 0), ibs2 - 16, ibs2)
    f1 = g.matmul_const((lr2, 0), emb, ffn)
    f1q = g.requant((f1, 0), 2.5 / math.sqrt(emb) / 127.0, dense_output_bitsize(emb))
    act = g.gelu((f1q, 0)) if gelu else g.relu((f1q, 0))
    f2 = g.matmul_const((act, 0), ffn, emb)
    f2q = g.requant((f2, 0), 2.5 / math.sqrt(ffn) / 127.0, dense_output_bitsize(ffn))
    res2 = g.add2((f2q, 0), (res1, 0))
    return g.set_outputs([(res2, 0)])


def matmul_pair(seq, k, n, config, transpose_b=False):
    """MatMul of two INPUT tensors, + a third one, Requant, ReLU (layers/matrix_mul.rs with two Input operands, layers/add.rs without operand)"""
    g = GraphBuilder([seq * k, k * n, seq * n], config)
    mm = g.matmul2((-1, 0), (-1, 1), k, n, transpose_b)
    ad = g.add2((mm, 0), (-1, 2), 1, 3)
    r = g.requant((ad, 0), 1.0 / math.sqrt(k) / 127.0, dense_output_bitsize(k) + 2)
    g.relu((r, 0))
    return g


def qkv_two_outputs(seq, k, n, config):
    """QKV whose Q is a model output and whose K + 2 V is another: a three-output node feeding an output and a later node"""
    g = GraphBuilder([seq * k], config)
    q = g.qkv((-1, 0), k, n)
    ad = g.add2((q, 1), (q, 2), 1, 2)
    return g.set_outputs([(q, 0), (ad, 0)])


def mlp(num_dense, width, config, input_features=4, output_features=3):
    """zkml/assets/scripts/MLP/mlp.py:54-83: Linear(4,W)+ReLU, (num_dense-1) x [Linear(W,W)+ReLU], Linear(W,3)+ReLU;
    every Dense is followed by its Requant node (quantisation inserts it), node ids as Model::random_with_rng."""
    mb = ModelBuilder(input_features, config)
    mb.dense(width, input_features).relu()
    for _ in range(num_dense - 1):
        mb.dense(width, width).relu()
    mb.dense(output_features, width).relu()
    return mb


def gelu_only(n, config, in_scale=1.0 / 128.0):
    """the reference's own GELU proving test (zkml/src/layers/activation.rs:686-697: one Activation::Gelu over a small tensor) at `n` entries"""
    return ModelBuilder(n, config).gelu(in_scale)


def gelu_mlp(width, config, input_features=4, output_features=3, in_scale=1.0 / 128.0):
    """Linear(4, W) + GELU + Linear(W, 3): the feed-forward shape with the activation a transformer uses (layers/activation.rs Activation::Gelu)"""
    mb = ModelBuilder(input_features, config)
    mb.dense(width, input_features).gelu(in_scale)
    mb.dense(output_features, width)
    return mb


def seq_mlp(seq, width, config, input_features=4, output_features=3, layers=2, transpose_last=False, positional=False):
    """a per-token MLP over a [seq][features] activation: MatMul(+bias)+Requant+ReLU blocks (the Linear layers of a transformer
    block applied to every position; layers/matrix_mul.rs with a constant right matrix), the last one without bias"""
    mb = ModelBuilder((seq, input_features), config)
    if positional:  # a learned positional table added to the input (transformer/positional.rs -> Add with a static operand)
        mb.add_const(1, 1)
    mb.matmul(width).relu()
    for _ in range(layers - 1):
        mb.matmul(width).relu()
    mb.matmul(output_features, bias=False, transpose_b=transpose_last).relu()
    return mb


def softmax_only(heads, n, config, in_scale=1.0 / 127.0):
    """Softmax over [heads][n][n] scores (layers/transformer/softmax.rs); in_scale 8/127 needs the zero tables"""
    mb = ModelBuilder((heads, n, n), config)
    mb.softmax(in_scale=in_scale)
    return mb


def layernorm_mlp(seq, features, width, config, output_features=3):
    """LayerNorm -> Linear -> ReLU -> Linear over a [seq][features] activation: the normalisation in front of the feed-forward part of a
    transformer block (layers/transformer/layernorm.rs, layers/matrix_mul.rs)"""
    mb = ModelBuilder((seq, features), config)
    mb.layernorm().matmul(width).relu()
    mb.matmul(output_features, bias=False).relu()
    return mb


def token_mlp(seq, vocab, width, config, output_features=3, max_positions=0):
    """tokens -> Embeddings -> + positional table -> two Linear layers per token: the non-attention part of a small language model
    (layers/transformer/embeddings.rs, layers/add.rs, layers/matrix_mul.rs)"""
    mb = ModelBuilder((seq,), config)
    mb.embeddings(vocab, width)
    if max_positions:  # Positional::Learned: a longer table, its first `seq` rows added (the claim is lifted to the whole table)
        mb.positional(max_positions)
    else:
        mb.add_const(1, 1)
    mb.matmul(width).relu()
    mb.matmul(output_features, bias=False).relu()
    return mb


def seq_1m():
    """a sequence of 64 tokens through two 1024-wide Linear layers (MatMul + bias + Requant + ReLU) and a 1024 -> 3 head: 1.05 M
    parameters, 2^16-row lookups per block — the MatMul workload of DESIGN.md section 6"""
    return seq_mlp(64, 1024, config=5, layers=2)


def dense_4m():
    """BASELINE config 2: 'Dense 4M' = mlp.py --num-dense 5 --layer-width 1024 (4.21 M parameters)"""
    return mlp(5, 1024, config=2)


def dense_128():
    """BASELINE config 1: a single Dense 128 -> 128, no requant / relu (plumbing case)"""
    mb = ModelBuilder(128, config=1)
    mb.dense(128, 128, requant=False)
    return mb


def cnn(c1, c2, fc1, fc2, fc3, config, input_shape=(3, 32, 32), kernel=5):
    """zkml/assets/scripts/CNN/cifar-cnn.py:194-254: conv(3,c1,5) relu pool conv(c1,c2,5) relu pool flatten fc1 relu fc2
    relu fc3; quantisation inserts a Requant after every conv / dense."""
    mb = ModelBuilder(input_shape, config)
    mb.conv(c1, kernel).relu().maxpool()
    mb.conv(c2, kernel).relu().maxpool()
    mb.flatten()
    mb.dense(fc1).relu()
    mb.dense(fc2).relu()
    mb.dense(fc3)
    return mb


def cnn_264k():
    """BASELINE config 3: 'CNN 264k' = cifar-cnn.py --num-params 264000 -> c1=12, c2=33, fc1=247, fc2=173, fc3=10
    (scale = sqrt(264000 / 62006))"""
    return cnn(12, 33, 247, 173, 10, config=3)


def cnn_tiny(config=9):
    """a small CNN of the same structure for tests (8x8 input, kernel 3)"""
    mb = ModelBuilder((2, 16, 16), config)
    mb.conv(3, 3).relu().maxpool()
    mb.flatten()
    mb.dense(5).relu()
    mb.dense(3)
    return mb
