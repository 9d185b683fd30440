"""The reference's proof wire format: `rmp_serde::to_vec_named(&Proof)` (zkml/src/bin/bench.rs:399) rebuilt from the canonical
u64 stream of csrc/proof.h, and back. SURVEY.md §8(f2).

CANNOT BE VERIFIED HERE: the reference is pure Rust with un-vendored dependencies and no Rust toolchain exists in this
environment, so not one byte of a reference-serialised proof is available to compare with. What IS fixed by sources on disk:
every struct / field / variant name and their order (file:line below). What is RECALLED (rmp-serde 1.3.0 and the serde derives
of Plonky3 @ f37dc2a5, neither on disk) is isolated in `Conventions` so that a maintainer with a toolchain flips a switch
instead of rewriting the encoder:
  * `to_vec_named`: struct -> map keyed by field name; newtype struct -> its inner value; unit variant -> the variant name as a
    string; newtype / tuple / struct variant -> a 1-entry map {name: payload}; Option -> nil | payload; tuples and Vec -> array;
    integers in the smallest MessagePack encoding; `PhantomData` (serialize_unit_struct) -> an empty array;
  * `Goldilocks` derives Serialize on `{ value: u64 }` -> {"value": u64} (possibly non-canonical in the reference: SURVEY F5;
    always canonical here); `BinomialExtensionField` on `{ value: [F; 2] }` -> {"value": [c0, c1]};
  * `HashMap<NodeId, LayerProof>` iterates in arbitrary order in the reference: ascending NodeId here.
Reference quirk reproduced on purpose: `FieldType::Base(#[serde(skip)] Vec<F>)` (multilinear_extensions/src/mle.rs:137-139) —
a base-field trivial opening is serialised as the bare variant name "Base" WITHOUT its evaluations, so the wire format cannot
carry what `PCS::verify` needs for it; `from_rmp` therefore returns such tables empty and `stream_equal_modulo_skipped` exists.

Layout sources: zkml/src/iop/mod.rs:21-44 (Proof, TableProof), layers/mod.rs:128-150 (LayerProof), layers/dense.rs:57-68,
requant.rs:84-99, activation.rs:61-72, commit/same_poly.rs:42-46, convolution.rs:97-121, hadamard.rs:51-54, pooling.rs:61-75,
lookup/logup_gkr/structs.rs:305-319, lib.rs:45-48 (Claim), sumcheck/src/structs.rs:15-34, commit/context.rs:224-232,
mpcs/src/basefold/structure.rs:161-166,334-345, query_phase.rs:541-544,609-615,655-662,722-741,1080-1087,1251-1257,
mpcs/src/util/merkle_tree.rs:156-162, sum_check/classic.rs:168-174, classic/coeff.rs:39, poseidon/src/digest.rs:7."""
import struct

import numpy as np

MAGIC = 0x31464F4F52505044
L_DENSE, L_REQUANT, L_RELU, L_CONV, L_MAXPOOL, L_MATMUL, L_ADD, L_EMBED, L_POSITIONAL = 0, 1, 2, 3, 4, 6, 7, 8, 9
# nodes of a model graph: MatMul / Add of two inputs share the reference's MatMulProof / AddProof variants with the constant forms (on the way
# back from the wire format they come out as kinds 6 / 7); ConcatMatMul and QKV have their own
L_MATMUL2, L_ADD2, L_CONCAT_MATMUL, L_QKV, L_LAYERNORM, L_SOFTMAX, L_MHA = 10, 11, 12, 13, 14, 15, 16
L_GELU = 17  # Activation::Gelu: the reference's ActivationProof, as for a Relu (back from the wire format it is kind 2)


class Conventions:
    """the recalled (not on disk) parts of the format"""
    field_as_map = True          # Goldilocks -> {"value": u64}; False: a bare u64
    ext_as_map = True            # Ext2 -> {"value": [c0, c1]}; False: a bare [c0, c1]
    phantom_is_empty_array = True  # PhantomData -> [] (serialize_unit_struct of rmp-serde); False: nil


# ---------------------------------------------------------------- canonical stream -> tree of Python values
class _Reader:
    def __init__(self, w):
        self.w, self.p = [int(x) for x in np.asarray(w, dtype=np.uint64)], 0

    def u(self):
        v = self.w[self.p]
        self.p += 1
        return v

    def e(self):
        return (self.u(), self.u())

    def ve(self):
        return [self.e() for _ in range(self.u())]

    def d(self):
        return [self.u() for _ in range(4)]

    def iop(self):
        point = self.ve()
        return {"point": point, "proofs": [self.ve() for _ in range(self.u())]}

    def claim(self):
        return {"point": self.ve(), "eval": self.e()}

    def logup(self):
        return {"sumcheck_proofs": [self.iop() for _ in range(self.u())], "round_evaluations": [self.ve() for _ in range(self.u())],
                "output_claims": [self.claim() for _ in range(self.u())], "circuit_outputs": [self.ve() for _ in range(self.u())],
                "is_table": self.u() != 0}

    def comm(self):
        return {"root": self.d(), "num_vars": self.u(), "is_base": self.u() != 0}

    def cq(self):
        ext = self.u() != 0
        pair = (self.e(), self.e()) if ext else (self.u(), self.u())
        return {"is_ext": ext, "pair": pair, "index": self.u(), "path": [self.d() for _ in range(self.u())]}

    def basefold(self):
        p = {"sumcheck_messages": [self.ve() for _ in range(self.u())], "roots": [self.d() for _ in range(self.u())], "final_message": self.ve()}
        p["queries"] = [{"index": self.u(), "oracle_query": [self.cq() for _ in range(self.u())], "commitments_query": [self.cq() for _ in range(self.u())]}
                        for _ in range(self.u())]
        p["sumcheck_proof"] = [self.ve() for _ in range(self.u())]
        tp = []
        for _ in range(self.u()):
            ext, n = self.u() != 0, self.u()
            tp.append({"is_ext": ext, "w": [self.u() for _ in range(2 * n if ext else n)]})
        p["trivial_proof"] = tp
        return p


def parse_stream(words):
    r = _Reader(words)
    assert r.u() == MAGIC, "not a canonical proof stream"
    steps = []
    for _ in range(r.u()):
        node, kind = r.u(), r.u()
        if kind == L_DENSE:
            lp = {"sumcheck": r.iop(), "bias_eval": r.e(), "individual_claims": r.ve()}
        elif kind == L_POSITIONAL:
            assert r.u() == 1
            lp = {"sub_matrix_evals": r.ve(), "left_eval": r.e(), "right_eval": r.e()}
        elif kind == L_EMBED:
            lp = {"sumcheck": r.iop(), "individual_claims": r.ve()}
        elif kind in (L_ADD, L_ADD2):
            lp = {"left_eval": r.e(), "right_eval": r.e()}
        elif kind in (L_MATMUL, L_MATMUL2):
            lp = {"sumcheck": r.iop(), "individual_claims": r.ve()}
            lp["bias_eval"] = r.e() if r.u() else None
        elif kind == L_CONCAT_MATMUL:
            lp = {"sumcheck_proof": r.iop(), "individual_claims": r.ve()}
        elif kind == L_QKV:
            lp = {"sumcheck": r.iop(), "aggregation_proof": {"sumcheck": r.iop(), "evals": r.ve()}, "pre_bias_evals": r.ve(), "individual_claims": r.ve()}
        elif kind == L_SOFTMAX:
            lp = {"logup_proofs": [r.logup() for _ in range(r.u())], "commitments": [r.comm() for _ in range(r.u())], "accumulation_proof": r.iop(), "mask_proof": r.iop(), "evaluations": r.ve()}
        elif kind == L_MHA:  # MhaProof {final_mul_proof, softmax_proof, qk_proof} (layers/transformer/mha.rs:122-128)
            lp = {"final_mul_proof": {"sumcheck_proof": r.iop(), "individual_claims": r.ve()}}
            lp["softmax_proof"] = {"logup_proofs": [r.logup() for _ in range(r.u())], "commitments": [r.comm() for _ in range(r.u())], "accumulation_proof": r.iop(), "mask_proof": r.iop(), "evaluations": r.ve()}
            lp["qk_proof"] = {"sumcheck_proof": r.iop(), "individual_claims": r.ve()}
        elif kind == L_LAYERNORM:
            lp = {"logup_proofs": [r.logup() for _ in range(r.u())], "commitments": [r.comm() for _ in range(r.u())], "accumulation_proof": r.iop(), "io_proof": r.iop(),
                  "input_proof": r.iop(), "acc_evals": r.ve(), "evaluations": r.ve(), "gamma_eval": r.e(), "beta_eval": r.e()}
        elif kind == L_REQUANT:
            lp = {"io_accumulation": r.iop(), "accumulation_evals": r.ve(), "clamping_lookup": r.logup(), "shifted_lookup": r.logup(),
                  "commitments": [r.comm() for _ in range(r.u())]}
        elif kind in (L_RELU, L_GELU):
            lp = {"io_accumulation": {"sumcheck": r.iop(), "evals": r.ve()}, "lookup": r.logup(), "commits": [r.comm() for _ in range(r.u())]}
        elif kind == L_CONV:
            lp = {"fft_proof": r.iop(), "fft_proof_weights": r.iop(), "fft_delegation_proof": [r.iop() for _ in range(r.u())],
                  "fft_delegation_proof_weights": [r.iop() for _ in range(r.u())], "ifft_proof": r.iop(),
                  "ifft_delegation_proof": [r.iop() for _ in range(r.u())], "hadamard_proof": r.iop(), "fft_claims": r.ve(), "fft_weight_claims": r.ve(),
                  "ifft_claims": r.ve(), "fft_delegation_claims": [r.ve() for _ in range(r.u())], "fft_delegation_weights_claims": [r.ve() for _ in range(r.u())],
                  "ifft_delegation_claims": [r.ve() for _ in range(r.u())], "partial_evals": r.ve(), "hadamard_clams": r.ve(), "bias_claim": r.e(),
                  "clearing_proof": {"sumcheck": r.iop(), "individual_claim": r.ve()}}
        elif kind == L_MAXPOOL:
            lp = {"sumcheck": r.iop(), "lookup": r.logup(), "zerocheck_evals": r.ve(), "variable_gap": r.u(), "commitments": [r.comm() for _ in range(r.u())]}
        else:
            raise ValueError(f"unknown layer kind {kind}")
        steps.append((node, kind, lp))
    tables = [{"multiplicity_commit": r.comm(), "lookup": r.logup()} for _ in range(r.u())]
    batch = r.basefold()
    trivial = [r.basefold() for _ in range(r.u())]
    assert r.p == len(r.w), "trailing words"
    return {"steps": steps, "table_proofs": tables, "batch_proof": batch, "trivial_proofs": trivial}


# ---------------------------------------------------------------- tree -> the serde data model (plain dict / list / int / str / None / bool)
class Phantom:
    pass


def _f(v, c):
    return {"value": v} if c.field_as_map else v


def _e(x, c):
    pair = [_f(x[0], c), _f(x[1], c)]
    return {"value": pair} if c.ext_as_map else pair


def _ve(v, c):
    return [_e(x, c) for x in v]


def _digest(d, c):
    return [_f(x, c) for x in d]  # Digest(pub [F; 4]): newtype struct -> the array


def _iop(p, c):
    return {"point": _ve(p["point"], c), "proofs": [{"evaluations": _ve(m, c)} for m in p["proofs"]]}


def _logup(p, c):
    return {"sumcheck_proofs": [_iop(s, c) for s in p["sumcheck_proofs"]], "round_evaluations": [_ve(r, c) for r in p["round_evaluations"]],
            "output_claims": [{"point": _ve(q["point"], c), "eval": _e(q["eval"], c)} for q in p["output_claims"]],
            "circuit_outputs": [_ve(o, c) for o in p["circuit_outputs"]], "proof_type": "Table" if p["is_table"] else "Lookup"}


def _comm(k, c):
    return {"root": _digest(k["root"], c), "num_vars": k["num_vars"], "is_base": k["is_base"], "num_polys": 1}  # Option::Some -> the payload


def _cq(q, c):
    pair = {"Ext": [_e(q["pair"][0], c), _e(q["pair"][1], c)]} if q["is_ext"] else {"Base": [_f(q["pair"][0], c), _f(q["pair"][1], c)]}
    return {"query": {"codepoints": pair, "index": q["index"]}, "merkle_path": {"inner": [_digest(d, c) for d in q["path"]], "_phantom": Phantom()}}


def _basefold(p, c, simple_batch=False):
    trivial = bool(p["trivial_proof"])
    if trivial or not p["queries"]:
        qr = {"Single": {"inner": []}}  # BasefoldProof::trivial (structure.rs:352-363)
    elif simple_batch:  # PCS::simple_batch_open (basefold.rs:846-860): one row pair (a pair per polynomial) and ONE Merkle path per query
        def row(q):     # SimpleBatchCommitmentSingleQueryResultWithMerklePath {query: {leaves: SimpleBatchLeavesPair, index}, merkle_path} (query_phase.rs:1328-1383, 556-566)
            cs = q["commitments_query"]
            assert all(not x["path"] and x["index"] == cs[0]["index"] and x["is_ext"] == cs[0]["is_ext"] for x in cs[1:]), "not a simple-batch opening"
            leaves = {"Ext": [[_e(x["pair"][0], c), _e(x["pair"][1], c)] for x in cs]} if cs[0]["is_ext"] else {"Base": [[_f(x["pair"][0], c), _f(x["pair"][1], c)] for x in cs]}
            return {"query": {"leaves": leaves, "index": cs[0]["index"]}, "merkle_path": {"inner": [_digest(d, c) for d in cs[0]["path"]], "_phantom": Phantom()}}
        qr = {"SimpleBatched": {"inner": [[q["index"], {"oracle_query": {"inner": [_cq(x, c) for x in q["oracle_query"]]}, "commitment_query": row(q)}] for q in p["queries"]]}}
    elif not p["sumcheck_proof"]:  # PCS::open of one polynomial (basefold.rs:532-544): no batch sumcheck, one commitment pair per query
        qr = {"Single": {"inner": [[q["index"], {"oracle_query": {"inner": [_cq(x, c) for x in q["oracle_query"]]},
                                                 "commitment_query": _cq(q["commitments_query"][0], c)}] for q in p["queries"]]}}
    else:
        qr = {"Batched": {"inner": [[q["index"], {"oracle_query": {"inner": [_cq(x, c) for x in q["oracle_query"]]},
                                                  "commitments_query": {"inner": [_cq(x, c) for x in q["commitments_query"]]}}] for q in p["queries"]]}}
    sp = None if not p["sumcheck_proof"] else {"rounds": [{"Ext": _ve(m, c)} for m in p["sumcheck_proof"]], "phantom": Phantom()}
    tp = [({"Ext": _ve([(m["w"][2 * i], m["w"][2 * i + 1]) for i in range(len(m["w"]) // 2)], c)} if m["is_ext"] else "Base") for m in p["trivial_proof"]]
    return {"sumcheck_messages": [_ve(m, c) for m in p["sumcheck_messages"]], "roots": [_digest(d, c) for d in p["roots"]], "final_message": _ve(p["final_message"], c),
            "query_result_with_merkle_path": qr, "sumcheck_proof": sp, "trivial_proof": tp}


def to_serde_model(tree, conv=Conventions):
    c = conv
    steps = {}
    for node, kind, lp in sorted(tree["steps"], key=lambda s: s[0]):
        if kind == L_DENSE:
            v = {"Dense": {"sumcheck": _iop(lp["sumcheck"], c), "bias_eval": _e(lp["bias_eval"], c), "individual_claims": _ve(lp["individual_claims"], c)}}
        elif kind == L_POSITIONAL:  # PositionalProof {proofs: Vec<SinglePositionalProof {sub_matrix_evals, add_proof}>} (transformer/positional.rs:45-61)
            v = {"Positional": {"proofs": [{"sub_matrix_evals": _ve(lp["sub_matrix_evals"], c),
                                            "add_proof": {"left_eval": _e(lp["left_eval"], c), "right_eval": _e(lp["right_eval"], c)}}]}}
        elif kind == L_EMBED:  # EmbeddingsProof {sumcheck, individual_claims} (layers/transformer/embeddings.rs:60-67)
            v = {"Embeddings": {"sumcheck": _iop(lp["sumcheck"], c), "individual_claims": _ve(lp["individual_claims"], c)}}
        elif kind in (L_ADD, L_ADD2):  # AddProof {left_eval, right_eval} (layers/add.rs:59-63)
            v = {"Add": {"left_eval": _e(lp["left_eval"], c), "right_eval": _e(lp["right_eval"], c)}}
        elif kind == L_CONCAT_MATMUL:  # ConcatMatMulProof {sumcheck_proof, individual_claims} (layers/concat_matmul.rs:365-373)
            v = {"ConcatMatMul": {"sumcheck_proof": _iop(lp["sumcheck_proof"], c), "individual_claims": _ve(lp["individual_claims"], c)}}
        elif kind == L_QKV:  # QKVProof {sumcheck, aggregation_proof, pre_bias_evals, individual_claims: [(E, E); 3]} (layers/transformer/qkv.rs:63-83)
            ic = _ve(lp["individual_claims"], c)
            v = {"QKV": {"sumcheck": _iop(lp["sumcheck"], c),
                         "aggregation_proof": {"sumcheck": _iop(lp["aggregation_proof"]["sumcheck"], c), "evals": _ve(lp["aggregation_proof"]["evals"], c)},
                         "pre_bias_evals": _ve(lp["pre_bias_evals"], c), "individual_claims": [[ic[2 * q], ic[2 * q + 1]] for q in range(3)]}}
        elif kind in (L_MATMUL, L_MATMUL2):  # MatMulProof {sumcheck, individual_claims, bias_eval: Option<E>} (layers/matrix_mul.rs:153-161)
            v = {"MatMul": {"sumcheck": _iop(lp["sumcheck"], c), "individual_claims": _ve(lp["individual_claims"], c),
                            "bias_eval": None if lp["bias_eval"] is None else _e(lp["bias_eval"], c)}}
        elif kind == L_SOFTMAX:  # SoftmaxProof (layers/transformer/softmax.rs:102-117)
            v = {"Softmax": {"logup_proofs": [_logup(x, c) for x in lp["logup_proofs"]], "commitments": [_comm(k, c) for k in lp["commitments"]],
                             "accumulation_proof": _iop(lp["accumulation_proof"], c), "mask_proof": _iop(lp["mask_proof"], c), "evaluations": _ve(lp["evaluations"], c)}}
        elif kind == L_MHA:  # MhaProof {final_mul_proof: ConcatMatMulProof, softmax_proof: SoftmaxProof, qk_proof: ConcatMatMulProof} (transformer/mha.rs:122-128)
            sp = lp["softmax_proof"]
            cm = lambda q: {"sumcheck_proof": _iop(q["sumcheck_proof"], c), "individual_claims": _ve(q["individual_claims"], c)}
            v = {"Mha": {"final_mul_proof": cm(lp["final_mul_proof"]),
                         "softmax_proof": {"logup_proofs": [_logup(x, c) for x in sp["logup_proofs"]], "commitments": [_comm(k, c) for k in sp["commitments"]],
                                           "accumulation_proof": _iop(sp["accumulation_proof"], c), "mask_proof": _iop(sp["mask_proof"], c), "evaluations": _ve(sp["evaluations"], c)},
                         "qk_proof": cm(lp["qk_proof"])}}
        elif kind == L_LAYERNORM:  # LayerNormProof (layers/transformer/layernorm.rs:644-667), fields in declaration order
            v = {"LayerNorm": {"logup_proofs": [_logup(x, c) for x in lp["logup_proofs"]], "commitments": [_comm(k, c) for k in lp["commitments"]],
                               "accumulation_proof": _iop(lp["accumulation_proof"], c), "io_proof": _iop(lp["io_proof"], c), "input_proof": _iop(lp["input_proof"], c),
                               "acc_evals": _ve(lp["acc_evals"], c), "evaluations": _ve(lp["evaluations"], c), "gamma_eval": _e(lp["gamma_eval"], c), "beta_eval": _e(lp["beta_eval"], c)}}
        elif kind == L_REQUANT:
            v = {"Requant": {"io_accumulation": _iop(lp["io_accumulation"], c), "accumulation_evals": _ve(lp["accumulation_evals"], c),
                             "clamping_lookup": _logup(lp["clamping_lookup"], c), "shifted_lookup": _logup(lp["shifted_lookup"], c),
                             "commitments": [_comm(k, c) for k in lp["commitments"]]}}
        elif kind in (L_RELU, L_GELU):
            v = {"Activation": {"io_accumulation": {"sumcheck": _iop(lp["io_accumulation"]["sumcheck"], c), "evals": _ve(lp["io_accumulation"]["evals"], c)},
                                "lookup": _logup(lp["lookup"], c), "commits": [_comm(k, c) for k in lp["commits"]]}}
        elif kind == L_CONV:
            names_iop = ("fft_proof", "fft_proof_weights", "ifft_proof", "hadamard_proof")
            names_viop = ("fft_delegation_proof", "fft_delegation_proof_weights", "ifft_delegation_proof")
            names_ve = ("fft_claims", "fft_weight_claims", "ifft_claims", "partial_evals", "hadamard_clams")
            names_vve = ("fft_delegation_claims", "fft_delegation_weights_claims", "ifft_delegation_claims")
            order = ("fft_proof", "fft_proof_weights", "fft_delegation_proof", "fft_delegation_proof_weights", "ifft_proof", "ifft_delegation_proof", "hadamard_proof",
                     "fft_claims", "fft_weight_claims", "ifft_claims", "fft_delegation_claims", "fft_delegation_weights_claims", "ifft_delegation_claims", "partial_evals",
                     "hadamard_clams", "bias_claim", "clearing_proof")
            body = {}
            for k in order:
                if k in names_iop:
                    body[k] = _iop(lp[k], c)
                elif k in names_viop:
                    body[k] = [_iop(x, c) for x in lp[k]]
                elif k in names_ve:
                    body[k] = _ve(lp[k], c)
                elif k in names_vve:
                    body[k] = [_ve(x, c) for x in lp[k]]
                elif k == "bias_claim":
                    body[k] = _e(lp[k], c)
                else:
                    body[k] = {"sumcheck": _iop(lp[k]["sumcheck"], c), "individual_claim": _ve(lp[k]["individual_claim"], c)}
            v = {"Convolution": body}
        else:
            v = {"Pooling": {"sumcheck": _iop(lp["sumcheck"], c), "lookup": _logup(lp["lookup"], c), "zerocheck_evals": _ve(lp["zerocheck_evals"], c),
                             "variable_gap": lp["variable_gap"], "commitments": [_comm(k, c) for k in lp["commitments"]]}}
        steps[node] = v
    return {"steps": steps, "table_proofs": [{"multiplicity_commit": _comm(t["multiplicity_commit"], c), "lookup": _logup(t["lookup"], c)} for t in tree["table_proofs"]],
            "commit": {"batch_proof": _basefold(tree["batch_proof"], c), "trivial_proofs": [_basefold(t, c) for t in tree["trivial_proofs"]]}}


# ---------------------------------------------------------------- MessagePack, the subset rmp-serde emits
def _pack(o, out, conv):
    if o is None:
        out.append(b"\xc0")
    elif o is True:
        out.append(b"\xc3")
    elif o is False:
        out.append(b"\xc2")
    elif isinstance(o, Phantom):
        out.append(b"\x90" if conv.phantom_is_empty_array else b"\xc0")
    elif isinstance(o, int):
        if o < 0x80:
            out.append(struct.pack("B", o))
        elif o < 1 << 8:
            out.append(b"\xcc" + struct.pack("B", o))
        elif o < 1 << 16:
            out.append(b"\xcd" + struct.pack(">H", o))
        elif o < 1 << 32:
            out.append(b"\xce" + struct.pack(">I", o))
        else:
            out.append(b"\xcf" + struct.pack(">Q", o))
    elif isinstance(o, str):
        b = o.encode()
        out.append((struct.pack("B", 0xA0 | len(b)) if len(b) < 32 else b"\xd9" + struct.pack("B", len(b))) + b)
    elif isinstance(o, list):
        n = len(o)
        out.append(struct.pack("B", 0x90 | n) if n < 16 else b"\xdc" + struct.pack(">H", n) if n < 1 << 16 else b"\xdd" + struct.pack(">I", n))
        for x in o:
            _pack(x, out, conv)
    elif isinstance(o, dict):
        n = len(o)
        out.append(struct.pack("B", 0x80 | n) if n < 16 else b"\xde" + struct.pack(">H", n) if n < 1 << 16 else b"\xdf" + struct.pack(">I", n))
        for k, v in o.items():
            _pack(k, out, conv)
            _pack(v, out, conv)
    else:
        raise TypeError(type(o))


def to_rmp(proof_words, conv=Conventions):
    """canonical stream (np.uint64) -> bytes of rmp_serde::to_vec_named(&Proof)"""
    out = []
    _pack(to_serde_model(parse_stream(proof_words), conv), out, conv)
    return b"".join(out)


def pcs_proof_to_rmp(proof_words, simple_batch=False, conv=Conventions):
    """stream of ONE Basefold proof (dp_pcs_open / dp_pcs_batch_open{,_evals} / dp_pcs_simple_batch_open) -> bytes of
    rmp_serde::to_vec_named(&BasefoldProof) (structure.rs:334-345). The canonical stream carries no variant tag: Batched is recognised by its
    batch sumcheck, Single by its absence; a simple-batch opening of ONE polynomial has the same stream as a Single one, so the caller says
    which of the two it holds (`simple_batch`)."""
    r = _Reader(proof_words)
    p = r.basefold()
    assert r.p == len(r.w), "trailing words"
    out = []
    _pack(_basefold(p, conv, simple_batch), out, conv)
    return b"".join(out)


def pcs_proof_from_rmp(data, conv=Conventions):
    """bytes of rmp_serde::to_vec_named(&BasefoldProof) -> the canonical stream of that proof"""
    model, end = _unpack(data, 0)
    assert end == len(data), "trailing bytes"
    w = _Writer(conv)
    w.basefold(model)
    return np.array(w.w, dtype=np.uint64)


def _unpack(b, p):
    t = b[p]
    if t < 0x80:
        return t, p + 1
    if t == 0xC0:
        return None, p + 1
    if t in (0xC2, 0xC3):
        return t == 0xC3, p + 1
    if t in (0xCC, 0xCD, 0xCE, 0xCF):
        n = {0xCC: 1, 0xCD: 2, 0xCE: 4, 0xCF: 8}[t]
        return int.from_bytes(b[p + 1:p + 1 + n], "big"), p + 1 + n
    if 0xA0 <= t < 0xC0 or t == 0xD9:
        n, q = (t & 0x1F, p + 1) if t != 0xD9 else (b[p + 1], p + 2)
        return b[q:q + n].decode(), q + n
    if 0x90 <= t < 0xA0 or t in (0xDC, 0xDD):
        n, q = (t & 0xF, p + 1) if t < 0xA0 else (int.from_bytes(b[p + 1:p + 3], "big"), p + 3) if t == 0xDC else (int.from_bytes(b[p + 1:p + 5], "big"), p + 5)
        o = []
        for _ in range(n):
            v, q = _unpack(b, q)
            o.append(v)
        return o, q
    if 0x80 <= t < 0x90 or t in (0xDE, 0xDF):
        n, q = (t & 0xF, p + 1) if t < 0x90 else (int.from_bytes(b[p + 1:p + 3], "big"), p + 3) if t == 0xDE else (int.from_bytes(b[p + 1:p + 5], "big"), p + 5)
        o = {}
        for _ in range(n):
            k, q = _unpack(b, q)
            v, q = _unpack(b, q)
            o[k] = v
        return o, q
    raise ValueError(f"unexpected MessagePack type byte {t:#x}")


# ---------------------------------------------------------------- serde model -> canonical stream
class _Writer:
    def __init__(self, conv):
        self.w, self.c = [], conv

    def f(self, v):
        self.w.append(v["value"] if self.c.field_as_map else v)

    def e(self, x):
        pair = x["value"] if self.c.ext_as_map else x
        self.f(pair[0]); self.f(pair[1])

    def ve(self, v):
        self.w.append(len(v))
        for x in v:
            self.e(x)

    def d(self, d):
        for x in d:
            self.f(x)

    def iop(self, p):
        self.ve(p["point"]); self.w.append(len(p["proofs"]))
        for m in p["proofs"]:
            self.ve(m["evaluations"])

    def logup(self, p):
        self.w.append(len(p["sumcheck_proofs"]))
        for s in p["sumcheck_proofs"]:
            self.iop(s)
        self.w.append(len(p["round_evaluations"]))
        for r in p["round_evaluations"]:
            self.ve(r)
        self.w.append(len(p["output_claims"]))
        for q in p["output_claims"]:
            self.ve(q["point"]); self.e(q["eval"])
        self.w.append(len(p["circuit_outputs"]))
        for o in p["circuit_outputs"]:
            self.ve(o)
        self.w.append(1 if p["proof_type"] == "Table" else 0)

    def comm(self, k):
        self.d(k["root"]); self.w.append(k["num_vars"]); self.w.append(1 if k["is_base"] else 0)

    def cq(self, q):
        cp = q["query"]["codepoints"]
        if "Ext" in cp:
            self.w.append(1); self.e(cp["Ext"][0]); self.e(cp["Ext"][1])
        else:
            self.w.append(0); self.f(cp["Base"][0]); self.f(cp["Base"][1])
        self.w.append(q["query"]["index"]); self.w.append(len(q["merkle_path"]["inner"]))
        for d in q["merkle_path"]["inner"]:
            self.d(d)

    def basefold(self, p):
        self.w.append(len(p["sumcheck_messages"]))
        for m in p["sumcheck_messages"]:
            self.ve(m)
        self.w.append(len(p["roots"]))
        for d in p["roots"]:
            self.d(d)
        self.ve(p["final_message"])
        qr = p["query_result_with_merkle_path"]
        if "Batched" in qr:
            qs = qr["Batched"]["inner"]
        elif "SimpleBatched" in qr:  # one stream entry per polynomial, all with the row pair's index, the path on the first
            qs = []
            for idx, q in qr["SimpleBatched"]["inner"]:
                cq = q["commitment_query"]
                (kind, pairs), = cq["query"]["leaves"].items()
                ent = [{"query": {"codepoints": {kind: pr}, "index": cq["query"]["index"]}, "merkle_path": {"inner": cq["merkle_path"]["inner"] if k == 0 else []}} for k, pr in enumerate(pairs)]
                qs.append([idx, {"oracle_query": q["oracle_query"], "commitments_query": {"inner": ent}}])
        else:  # Single: the canonical stream keeps the one commitment pair as a list of one
            qs = [[idx, {"oracle_query": q["oracle_query"], "commitments_query": {"inner": [q["commitment_query"]]}}] for idx, q in qr["Single"]["inner"]]
        self.w.append(len(qs))
        for idx, q in qs:
            self.w.append(idx)
            self.w.append(len(q["oracle_query"]["inner"]))
            for x in q["oracle_query"]["inner"]:
                self.cq(x)
            self.w.append(len(q["commitments_query"]["inner"]))
            for x in q["commitments_query"]["inner"]:
                self.cq(x)
        rounds = p["sumcheck_proof"]["rounds"] if p["sumcheck_proof"] is not None else []
        self.w.append(len(rounds))
        for m in rounds:
            self.ve(m["Ext"])
        self.w.append(len(p["trivial_proof"]))
        for m in p["trivial_proof"]:
            if m == "Base":  # #[serde(skip)]: the evaluations did not travel
                self.w.append(0); self.w.append(0)
            else:
                self.w.append(1); self.w.append(len(m["Ext"]))
                for x in m["Ext"]:
                    self.e(x)


def from_rmp(data, conv=Conventions):
    """bytes of rmp_serde::to_vec_named(&Proof) -> canonical stream (np.uint64); base-field trivial openings come back empty"""
    model, end = _unpack(data, 0)
    assert end == len(data), "trailing bytes"
    w = _Writer(conv)
    w.w.append(MAGIC); w.w.append(len(model["steps"]))
    kinds = {"Dense": L_DENSE, "Requant": L_REQUANT, "Activation": L_RELU, "Convolution": L_CONV, "Pooling": L_MAXPOOL, "MatMul": L_MATMUL, "Add": L_ADD, "Embeddings": L_EMBED, "Positional": L_POSITIONAL,
             "ConcatMatMul": L_CONCAT_MATMUL, "QKV": L_QKV, "LayerNorm": L_LAYERNORM, "Softmax": L_SOFTMAX, "Mha": L_MHA}
    for node in sorted(model["steps"]):
        (name, lp), = model["steps"][node].items()
        w.w.append(node); w.w.append(kinds[name])
        if name == "Dense":
            w.iop(lp["sumcheck"]); w.e(lp["bias_eval"]); w.ve(lp["individual_claims"])
        elif name == "Positional":
            assert len(lp["proofs"]) == 1
            w.w.append(1); w.ve(lp["proofs"][0]["sub_matrix_evals"]); w.e(lp["proofs"][0]["add_proof"]["left_eval"]); w.e(lp["proofs"][0]["add_proof"]["right_eval"])
        elif name == "Embeddings":
            w.iop(lp["sumcheck"]); w.ve(lp["individual_claims"])
        elif name == "Add":
            w.e(lp["left_eval"]); w.e(lp["right_eval"])
        elif name == "ConcatMatMul":
            w.iop(lp["sumcheck_proof"]); w.ve(lp["individual_claims"])
        elif name == "QKV":
            w.iop(lp["sumcheck"]); w.iop(lp["aggregation_proof"]["sumcheck"]); w.ve(lp["aggregation_proof"]["evals"]); w.ve(lp["pre_bias_evals"])
            w.ve([x for pair in lp["individual_claims"] for x in pair])
        elif name == "MatMul":
            w.iop(lp["sumcheck"]); w.ve(lp["individual_claims"])
            w.w.append(0 if lp["bias_eval"] is None else 1)
            if lp["bias_eval"] is not None:
                w.e(lp["bias_eval"])
        elif name == "Mha":
            w.iop(lp["final_mul_proof"]["sumcheck_proof"]); w.ve(lp["final_mul_proof"]["individual_claims"])
            sp = lp["softmax_proof"]
            w.w.append(len(sp["logup_proofs"]))
            for x in sp["logup_proofs"]:
                w.logup(x)
            w.w.append(len(sp["commitments"]))
            for k in sp["commitments"]:
                w.comm(k)
            w.iop(sp["accumulation_proof"]); w.iop(sp["mask_proof"]); w.ve(sp["evaluations"])
            w.iop(lp["qk_proof"]["sumcheck_proof"]); w.ve(lp["qk_proof"]["individual_claims"])
        elif name == "Softmax":
            w.w.append(len(lp["logup_proofs"]))
            for x in lp["logup_proofs"]:
                w.logup(x)
            w.w.append(len(lp["commitments"]))
            for k in lp["commitments"]:
                w.comm(k)
            w.iop(lp["accumulation_proof"]); w.iop(lp["mask_proof"]); w.ve(lp["evaluations"])
        elif name == "LayerNorm":
            w.w.append(len(lp["logup_proofs"]))
            for x in lp["logup_proofs"]:
                w.logup(x)
            w.w.append(len(lp["commitments"]))
            for k in lp["commitments"]:
                w.comm(k)
            w.iop(lp["accumulation_proof"]); w.iop(lp["io_proof"]); w.iop(lp["input_proof"]); w.ve(lp["acc_evals"]); w.ve(lp["evaluations"]); w.e(lp["gamma_eval"]); w.e(lp["beta_eval"])
        elif name == "Requant":
            w.iop(lp["io_accumulation"]); w.ve(lp["accumulation_evals"]); w.logup(lp["clamping_lookup"]); w.logup(lp["shifted_lookup"])
            w.w.append(len(lp["commitments"]))
            for k in lp["commitments"]:
                w.comm(k)
        elif name == "Activation":
            w.iop(lp["io_accumulation"]["sumcheck"]); w.ve(lp["io_accumulation"]["evals"]); w.logup(lp["lookup"])
            w.w.append(len(lp["commits"]))
            for k in lp["commits"]:
                w.comm(k)
        elif name == "Convolution":
            w.iop(lp["fft_proof"]); w.iop(lp["fft_proof_weights"])
            for k in ("fft_delegation_proof", "fft_delegation_proof_weights"):
                w.w.append(len(lp[k]))
                for x in lp[k]:
                    w.iop(x)
            w.iop(lp["ifft_proof"]); w.w.append(len(lp["ifft_delegation_proof"]))
            for x in lp["ifft_delegation_proof"]:
                w.iop(x)
            w.iop(lp["hadamard_proof"]); w.ve(lp["fft_claims"]); w.ve(lp["fft_weight_claims"]); w.ve(lp["ifft_claims"])
            for k in ("fft_delegation_claims", "fft_delegation_weights_claims", "ifft_delegation_claims"):
                w.w.append(len(lp[k]))
                for x in lp[k]:
                    w.ve(x)
            w.ve(lp["partial_evals"]); w.ve(lp["hadamard_clams"]); w.e(lp["bias_claim"])
            w.iop(lp["clearing_proof"]["sumcheck"]); w.ve(lp["clearing_proof"]["individual_claim"])
        else:
            w.iop(lp["sumcheck"]); w.logup(lp["lookup"]); w.ve(lp["zerocheck_evals"]); w.w.append(lp["variable_gap"])
            w.w.append(len(lp["commitments"]))
            for k in lp["commitments"]:
                w.comm(k)
    w.w.append(len(model["table_proofs"]))
    for t in model["table_proofs"]:
        w.comm(t["multiplicity_commit"]); w.logup(t["lookup"])
    w.basefold(model["commit"]["batch_proof"])
    w.w.append(len(model["commit"]["trivial_proofs"]))
    for t in model["commit"]["trivial_proofs"]:
        w.basefold(t)
    return np.array(w.w, dtype=np.uint64)


def strip_skipped(proof_words):
    """the canonical stream with every base-field trivial opening emptied: what survives the reference's wire format"""
    return from_rmp(to_rmp(proof_words))


def stream_equal_modulo_skipped(a, b):
    return bool(np.array_equal(strip_skipped(a), strip_skipped(b)))
