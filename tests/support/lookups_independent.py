"""A THIRD writing, in numpy, of what the lookup argument of a proof is fed with: every lookup TABLE (its columns, its multiplicities) and every lookup
WITNESS (the columns that enter logup-GKR and the committed ones) of one inference — next to the product's host code (csrc/zkml.h `table_columns`,
`witness_host`) and the oracle's (oracle/zkml.hpp `instantiate_witness_ctx`), which were typed by the same hand from the same reading of the reference
(126 shared lines). Written from the reference's text, not from either C++ copy:

  tables     zkml/src/lookup/context.rs:158-296 (TableType::get_merged_table_column / generate_lookup_table), :55-72 (the derive(Ord) order of the kinds)
  counting   lookup/context.rs:631-756 (generate_lookup_witnesses: element counts keyed by in + out * 2^32, multiplicity = count / repetitions of the row)
  Requant    layers/requant.rs:208-345      Relu / GELU  layers/activation.rs:238-318, 555-588      MaxPool  layers/pooling.rs:206-262, 686-767
  LayerNorm  layers/transformer/layernorm.rs:1103-1218      Softmax / Mha  layers/transformer/softmax.rs:890-1066, mha.rs:706-719

The arithmetic of the quantised operators (requant shift, the f32 look-up functions) comes from deep-prove_amd/models.py, itself a numpy restatement that the
inference tests compare with both C++ copies. TEST INFRASTRUCTURE: nothing in the product imports this file."""
import numpy as np

import deep_prove_amd.models as M

P = 0xFFFFFFFF00000001
SEP = 1 << 32
K_RELU, K_GELU, K_RANGE, K_CLAMP, K_SOFTMAX, K_ERROR, K_ZERO, K_INVSQRT = range(8)


def field(v):
    """i64 column -> canonical Goldilocks words (quantization/mod.rs:210-220: a negative v is p - |v|)"""
    v = np.asarray(v, dtype=np.int64)
    return np.where(v < 0, (v.astype(np.int64) + np.int64(P - (1 << 64))).astype(np.uint64), v.astype(np.uint64))


def tkey(t):
    """derive(Ord) of TableType (context.rs:55-72): the kind, then the payload in field order — SoftmaxTableData (float_bits, table_size, bkm),
    InverseSQRTTableData (eps_bits, range_check_bits), GELUQuantData (multiplier, ..), Clamping(size), ZeroTable(size), ErrorTable(.., error)"""
    return (t["kind"], t.get("aux", 0), t.get("size", 0), t.get("aux2", 0))


def table_columns(t):
    """the columns of one table in row order (context.rs:158-296)"""
    k = t["kind"]
    if k == K_RELU:
        i = np.arange(-128, 128, dtype=np.int64)
        return [i, np.maximum(i, 0)]
    if k == K_RANGE:
        return [np.arange(256, dtype=np.int64)]
    if k == K_CLAMP:
        i = np.arange(-(1 << (t["size"] - 1)), 1 << (t["size"] - 1), dtype=np.int64)
        return [i, np.clip(i, -127, 127)]
    if k == K_GELU:
        i = np.arange(-(1 << (t["size"] - 1)), 1 << (t["size"] - 1), dtype=np.int64)  # min .. max, max excluded (activation.rs:579-581)
        return [i, np.array([M.gelu_table_output(int(v)) for v in i], dtype=np.int64)]
    if k == K_SOFTMAX:
        j = np.arange(1 << t["size"], dtype=np.int64)
        l = dict(bkm=t["aux2"], temp_bits=t["aux"])
        return [j, np.array([M.softmax_table_output(l, int(v)) for v in j], dtype=np.int64)]
    if k == K_ERROR:  # one - error ..= one + error, cut or zero padded to 2^ceil_log2(2 error) rows (context.rs:248-264)
        n = 1 << max(0, (2 * t["aux2"] - 1).bit_length())
        v = np.arange(4096 - t["aux2"], 4096 + t["aux2"] + 1, dtype=np.int64)[:n]
        return [np.concatenate([v, np.zeros(n - v.size, dtype=np.int64)])]
    if k == K_ZERO:
        i = np.arange(1 << t["size"], dtype=np.int64)
        return [i, (i == 0).astype(np.int64)]
    if k == K_INVSQRT:
        i = np.arange(-(1 << 14), 1 << 14, dtype=np.int64)
        return [i, M.inv_sqrt_table_output(t["aux"], t["size"], i)]
    raise ValueError(k)


def merged(cols):
    return cols[0] if len(cols) == 1 else cols[0] + cols[1] * SEP


class Collector:
    def __init__(self):
        self.lookups, self.counts, self.tables = [], {}, {}

    def add(self, node, which, t, lookup_cols, committed=None):
        """one lookup of a node: `lookup_cols` in instance-major order (cpi columns per instance), every instance counted into its table"""
        cpi = 1 if t["kind"] in (K_RANGE, K_ERROR) else 2
        lookup_cols = [np.asarray(c, dtype=np.int64) for c in lookup_cols]
        committed = lookup_cols if committed is None else [np.asarray(c, dtype=np.int64) for c in committed]
        self.tables[tkey(t)] = t
        cnt = self.counts.setdefault(tkey(t), {})
        for i in range(0, len(lookup_cols), cpi):
            keys = merged(lookup_cols[i:i + cpi])
            u, c = np.unique(keys, return_counts=True)
            for a, b in zip(u.tolist(), c.tolist()):
                cnt[a] = cnt.get(a, 0) + b
        self.lookups.append(dict(node=node, which=which, t=t, lookup=lookup_cols, committed=committed))

    def table_records(self):
        """tables in derive(Ord) order with the multiplicity column: count(row) / (number of rows that hold the same merged value), in the field"""
        out = []
        for key in sorted(self.tables):
            t = self.tables[key]
            cols = table_columns(t)
            m = merged(cols)
            u, rep = np.unique(m, return_counts=True)
            reps = dict(zip(u.tolist(), rep.tolist()))
            cnt = self.counts[key]
            assert set(cnt) <= set(reps), f"a lookup misses table {key}: {sorted(set(cnt) - set(reps))[:4]}"
            mult = np.array([cnt.get(v, 0) * pow(reps[v], P - 2, P) % P if reps[v] != 1 else cnt.get(v, 0) for v in m.tolist()], dtype=np.uint64)
            out.append(dict(t=t, cols=cols, mult=mult))
        return out


def requant_witness(col, node, l, x):
    shift = l["fp_scale"] + l["right_shift"]
    tmp = np.asarray(x, dtype=np.int64) * l["fixed_point_multiplier"] + (1 << (shift - 1))
    cin = tmp >> shift
    low = tmp & ((1 << shift) - 1)
    size = l["intermediate_bit_size"] + max(0, (l["fixed_point_multiplier"] - 1).bit_length()) - shift  # Requant::clamping_size (requant.rs)
    col.add(node, 0, dict(kind=K_CLAMP, size=size), [cin, np.clip(cin, -127, 127)])
    col.add(node, 1, dict(kind=K_RANGE), [(low >> (8 * j)) & 255 for j in range(shift // 8)])
    return np.clip(cin, -127, 127)


def softmax_witness(col, node, l, x):
    tr = {}
    out = M.softmax_apply(l, x, tr)
    C, R, K = l["shape"]
    t = dict(kind=K_SOFTMAX, size=l["table_size"], aux=l["temp_bits"], aux2=l["bkm"])
    col.add(node, 0, t, [tr["exp_in"], tr["exp_out"]])
    col.add(node, 1, dict(kind=K_RANGE), [tr["low"], tr["high"]])
    rows = out.reshape(C * R, K).sum(axis=1)  # the row sums are what is looked up in the error table; the SHIFT column is what is committed with it
    col.add(node, 2, dict(kind=K_ERROR, aux2=l["allowable_error"]), [rows], committed=[tr["shift"]])
    if l["zero_chunks"]:
        zc = []
        for z in range(l["zero_chunks"]):
            zc += [tr["zero_in"][z], tr["zero_out"][z]]
        col.add(node, 3, dict(kind=K_ZERO, size=l["zero_vars"]), zc)
    return out


def layernorm_witness(col, node, l, x):
    out, (lin, inv, rc) = M.layernorm_apply(l, x)
    rcb = l["range_check_bits"]
    col.add(node, 0, dict(kind=K_INVSQRT, size=rcb, aux=l["eps_bits"]), [lin, inv])
    nrc = (rcb - 1) // 8 + 1
    top = 1 << ((8 - rcb % 8) % 8)  # the top chunk is scaled up so that the range check of 8 bits bounds it by its own width (layernorm.rs:1150-1180)
    col.add(node, 1, dict(kind=K_RANGE), [((rc >> (8 * j)) & 255) * (top if j + 1 == nrc else 1) for j in range(nrc)])
    return out


def maxpool_witness(col, node, l, x):
    c, h, w = l["pin"]
    t = np.asarray(x, dtype=np.int64).reshape(c, h // 2, 2, w // 2, 2)
    out = t.max(axis=(2, 4))
    diffs = [(out - t[:, :, dy, :, dx]).reshape(-1) for dy, dx in ((0, 0), (1, 0), (0, 1), (1, 1))]  # Maxpool2D::compute_polys (pooling.rs:686-767)
    col.add(node, 0, dict(kind=K_RANGE), diffs, committed=diffs + [out.reshape(-1)])
    return out.reshape(-1)


def activation_witness(col, node, l, x):
    x = np.asarray(x, dtype=np.int64)
    if l["kind"] == M.L_RELU:
        out = np.maximum(x, 0)
        col.add(node, 0, dict(kind=K_RELU), [x, out])
    else:
        m = l["multiplier"]
        out = M.gelu_apply(l, x)
        col.add(node, 0, dict(kind=K_GELU, size=8 + max(0, (m - 1).bit_length()), aux2=m), [x * m, out])  # the first column is the SCALED input (activation.rs:262-275)
    return out


WITNESS = {M.L_REQUANT: requant_witness, M.L_RELU: activation_witness, M.L_GELU: activation_witness, M.L_SOFTMAX: softmax_witness,
           M.L_LAYERNORM: layernorm_witness, M.L_MAXPOOL: maxpool_witness}


def collect(mb, x):
    """walk a ModelBuilder (a chain) or GraphBuilder over the input and collect every lookup; the operators without lookups come from the builder's own
    numpy inference one node at a time"""
    col = Collector()
    if hasattr(mb, "nodes"):
        x = np.asarray(x, dtype=np.int64)
        offs = np.concatenate([[0], np.cumsum(mb.input_lens)])
        vals = {}
        get = lambda e: x[offs[e[1]]:offs[e[1] + 1]] if e[0] < 0 else vals[tuple(e)]  # noqa: E731
        for i, (l, edges) in enumerate(mb.nodes):
            k = l["kind"]
            if k in WITNESS:
                vals[(i, 0)] = WITNESS[k](col, i, l, get(edges[0]))
            elif k == M.L_MHA:  # Mha::gen_lookup_witness (mha.rs:706-719): its softmax over the products Q K^T, under the node's own id
                S, H, D = l["shape"]
                qh, kh, vh = (get(e).reshape(S, H, D).transpose(1, 0, 2) for e in edges)
                probs = softmax_witness(col, i, dict(l, shape=(H, S, S)), np.einsum("hsd,htd->hst", qh, kh).reshape(-1)).reshape(H, S, S)
                vals[(i, 0)] = np.einsum("hst,htd->hsd", probs, vh).transpose(1, 0, 2).reshape(-1)
            else:  # one node of the builder's own inference
                sub = type(mb).__new__(type(mb)); sub.__dict__.update(mb.__dict__)
                sub.nodes = [(l, [(-1, j) for j in range(len(edges))])]; sub.input_lens = [get(e).size for e in edges]
                sub.outputs = [(0, w) for w in range(3 if k == M.L_QKV else 1)]
                y = sub.run(np.concatenate([get(e) for e in edges]))
                if k == M.L_QKV:
                    for w, part in enumerate(np.split(y, 3)):
                        vals[(i, w)] = part
                else:
                    vals[(i, 0)] = y
        return col
    cur = np.asarray(x, dtype=np.int64)
    for i, l in enumerate(mb.layers):
        if l["kind"] in WITNESS:
            cur = WITNESS[l["kind"]](col, i, l, cur)
        else:
            sub = type(mb).__new__(type(mb)); sub.__dict__.update(mb.__dict__); sub.layers = [l]
            cur = sub.run(cur)
    return col
