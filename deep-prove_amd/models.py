"""Synthetic, already-quantised models for BASELINE.json's configs (the ONNX / float front-end of the reference is out
of scope). Tensors are SplitMix64 streams seeded per tensor: seed = 0xD33B0000 ^ (config << 32) ^ index,
values uniform in [-127, 127] (quantization MIN..MAX, zkml/src/quantization/mod.rs:28-29)."""
import math

import numpy as np

L_DENSE, L_REQUANT, L_RELU = 0, 1, 2
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


def dense_output_bitsize(ncols):
    """Dense::output_bitsize (zkml/src/layers/dense.rs:413-419)"""
    return 2 * (BIT_LEN - 1) + max(0, (ncols - 1).bit_length()) + 1


def next_pow2(x):
    return 1 << max(0, (x - 1).bit_length())


class ModelBuilder:
    def __init__(self, input_len, config=0):
        self.input_len = next_pow2(input_len)
        self.layers = []
        self.config = config
        self._tensor_index = 0
        self._cur = self.input_len

    def _tensor(self, n):
        t = quantised_tensor(self.config, self._tensor_index, n)
        self._tensor_index += 1
        return t

    def dense(self, out_features, in_features=None, requant=True, float_abs_max=None):
        """Dense (padded to powers of two; padding rows/cols are zero like Tensor::pad_next_power_of_two) + Requant"""
        in_features = in_features or self._cur
        r, c = next_pow2(out_features), next_pow2(in_features)
        assert c == self._cur
        w = np.zeros((r, c), dtype=np.int64)
        w[:out_features, :in_features] = self._tensor(out_features * in_features).reshape(out_features, in_features)
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

    def relu(self):
        self.layers.append(dict(kind=L_RELU))
        return self

    def blob(self):
        out = [self.input_len, len(self.layers)]
        parts = [np.array(out, dtype=np.int64)]
        for l in self.layers:
            if l["kind"] == L_DENSE:
                parts.append(np.array([L_DENSE, l["nrows"], l["ncols"]], dtype=np.int64))
                parts.append(l["weights"].reshape(-1))
                parts.append(l["bias"])
            elif l["kind"] == L_REQUANT:
                parts.append(np.array([L_REQUANT, l["right_shift"], l["fp_scale"], l["fixed_point_multiplier"],
                                       l["intermediate_bit_size"]], dtype=np.int64))
            else:
                parts.append(np.array([L_RELU], dtype=np.int64))
        return np.concatenate(parts)

    def input(self, index=1000):
        return quantised_tensor(self.config, index, self.input_len)

    def run(self, x):
        """quantised inference in numpy (Model::run semantics: Dense matvec + bias, Requant::apply, Relu::apply)"""
        cur = np.asarray(x, dtype=np.int64)
        for l in self.layers:
            if l["kind"] == L_DENSE:
                cur = l["weights"] @ cur + l["bias"]
            elif l["kind"] == L_REQUANT:
                sh = l["fp_scale"] + l["right_shift"]
                cur = np.clip((cur * l["fixed_point_multiplier"] + (1 << (sh - 1))) >> sh, -127, 127)
            else:
                cur = np.maximum(cur, 0)
        return cur


def mlp(num_dense, width, config, input_features=4, output_features=3):
    """zkml/assets/scripts/MLP/mlp.py:54-83: Linear(4,W)+ReLU, (num_dense-1) x [Linear(W,W)+ReLU], Linear(W,3)+ReLU;
    every Dense is followed by its Requant node (quantisation inserts it), node ids as Model::random_with_rng."""
    mb = ModelBuilder(input_features, config)
    mb.dense(width, input_features).relu()
    for _ in range(num_dense - 1):
        mb.dense(width, width).relu()
    mb.dense(output_features, width).relu()
    return mb


def dense_4m():
    """BASELINE config 2: 'Dense 4M' = mlp.py --num-dense 5 --layer-width 1024 (4.21 M parameters)"""
    return mlp(5, 1024, config=2)


def dense_128():
    """BASELINE config 1: a single Dense 128 -> 128, no requant / relu (plumbing case)"""
    mb = ModelBuilder(128, config=1)
    mb.dense(128, 128, requant=False)
    return mb
