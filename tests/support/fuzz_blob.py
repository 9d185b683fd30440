import sys; import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import deep_prove_amd as dpa
rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 1)
bases = [dpa.models.mlp(1, 8, config=3), dpa.models.cnn_tiny(), dpa.models.seq_mlp(4, 8, config=4, transpose_last=True, positional=True), dpa.models.token_mlp(4, 10, 8, config=5, max_positions=9),
         # graph blobs: edges, several input / output tensors, ConcatMatMul geometry, QKV
         dpa.models.attention_block(4, 8, 2, 4, config=6), dpa.models.matmul_pair(4, 8, 8, config=7, transpose_b=True), dpa.models.qkv_two_outputs(4, 8, 8, config=8),
         dpa.models.gelu_mlp(8, config=9)]  # Activation::Gelu: the multiplier word
ok = err = 0
for it in range(int(sys.argv[2]) if len(sys.argv) > 2 else 3000):
    mb = bases[it % len(bases)]
    b = mb.blob().copy(); x = mb.input()
    k = int(rng.integers(1, 4))
    for _ in range(k):
        pos = int(rng.integers(0, max(1, min(b.size, 64)))) if rng.random() < 0.7 else int(rng.integers(0, max(1, b.size)))
        mode = rng.integers(0, 5)
        if mode == 0: b[pos] = int(rng.integers(-5, 70))
        elif mode == 1: b[pos] = int(rng.integers(-2**62, 2**62))
        elif mode == 2: b[pos] ^= 1 << int(rng.integers(0, 63))
        elif mode == 3: b = b[:int(rng.integers(1, max(2, b.size)))]
        else: b[pos] = [0, -1, 2**31, 2**32, 2**40, 2**63 - 1][int(rng.integers(0, 6))]
        if b.size == 0: break
    try:
        dpa.infer_host(b, x); ok += 1
    except dpa.DeepProveError:
        err += 1
print("fuzz done: accepted", ok, "rejected", err)
