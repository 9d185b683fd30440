"""Host logic of the product (transcript order, claim routing, proof assembly, verifier) checked WITHOUT a GPU: the
orchestrator of deep-prove_amd/csrc runs over the CPU test double of tests/support/cpu_dev.hpp and is byte-compared with
the oracle; the product verifier must accept both proofs and reject tampered ones."""
import os
import subprocess

import pytest


def run(binary, *args, env=None):
    import os
    e = dict(os.environ)
    e.update(env or {})
    return subprocess.run([binary, *map(str, args)], capture_output=True, text=True, timeout=600, env=e)


@pytest.mark.parametrize("width,seed", [(8, 5), (16, 7), (64, 1)])
def test_proof_stream_identical_to_oracle_and_accepted(hostlogic_bin, width, seed):
    r = run(hostlogic_bin, width, seed)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "identical=1" in r.stdout
    assert "verify(oracle): ACCEPT" in r.stdout and "verify(product): ACCEPT" in r.stdout


@pytest.mark.parametrize("args", [(16, 7), (64, 1), ("cnn", 4)])
def test_device_side_fiat_shamir_contract(hostlogic_bin, args):
    """Dev::sc_tail (the persistent sumcheck kernel runs the transcript rounds itself and hands the sponge back): with the
    CPU double taking the tails of small sumchecks and declining the others, the host orchestrator still produces the
    oracle's stream byte for byte"""
    r = run(hostlogic_bin, *args, env={"DP_DOUBLE_DEVICE_FS": "1"})
    assert r.returncode == 0, r.stdout + r.stderr
    assert "identical=1" in r.stdout
    taken = int(r.stdout.split("sc_tail: ")[1].split()[0])
    declined = int(r.stdout.split("taken by the double, ")[1].split()[0])
    assert taken > 10 and declined > 0, r.stdout
    assert "verify(oracle): ACCEPT" in r.stdout and "verify(product): ACCEPT" in r.stdout


@pytest.mark.parametrize("args,fs", [((16, 7), "0"), ((64, 1), "1"), (("cnn", 4), "1")])
def test_device_side_logup_contract(hostlogic_bin, args, fs):
    """Dev::logup_tail (all layers of a logup-GKR proof with the transcript on the device, alone or on top of sc_tail)"""
    r = run(hostlogic_bin, *args, env={"DP_DOUBLE_DEVICE_LOGUP": "1", "DP_DOUBLE_DEVICE_FS": fs})
    assert r.returncode == 0, r.stdout + r.stderr
    assert "identical=1" in r.stdout
    assert int(r.stdout.split("logup_tail: ")[1].split()[0]) >= 6, r.stdout
    assert "verify(oracle): ACCEPT" in r.stdout and "verify(product): ACCEPT" in r.stdout


@pytest.mark.parametrize("args,fs", [((16, 7), "0"), ((64, 1), "1"), (("cnn", 4), "1")])
def test_device_side_whole_logup_contract(hostlogic_bin, args, fs):
    """Dev::logup_full (a whole logup-GKR batch proof — trees, outputs, initial challenges, layers, column claims — with the
    transcript on the device)"""
    r = run(hostlogic_bin, *args, env={"DP_DOUBLE_DEVICE_LOGUP": "2", "DP_DOUBLE_DEVICE_FS": fs})
    assert r.returncode == 0, r.stdout + r.stderr
    assert "identical=1" in r.stdout
    assert int(r.stdout.split("logup_full: ")[1].split()[0]) >= 6, r.stdout
    assert "verify(oracle): ACCEPT" in r.stdout and "verify(product): ACCEPT" in r.stdout


@pytest.mark.parametrize("args", [(16, 7), (64, 1), ("cnn", 4)])
def test_device_side_classic_sumcheck_contract(hostlogic_bin, args):
    """Dev::classic_tail (the last rounds of the batch-opening sumcheck of pcs_batch_open with the transcript on the device),
    together with the other device-side transcript contracts"""
    r = run(hostlogic_bin, *args, env={"DP_DOUBLE_DEVICE_CLASSIC": "1", "DP_DOUBLE_DEVICE_DENSE": "1", "DP_DOUBLE_DEVICE_EQSUM": "1", "DP_DOUBLE_DEVICE_COMMIT": "1", "DP_DOUBLE_DEVICE_LOGUP": "2", "DP_DOUBLE_DEVICE_FS": "1"})
    assert r.returncode == 0, r.stdout + r.stderr
    assert "identical=1" in r.stdout
    assert int(r.stdout.split("classic_tail: ")[1].split()[0]) >= 1, r.stdout
    assert int(r.stdout.split("dense_tail: ")[1].split()[0]) >= 1, r.stdout  # Dev::dense_tail (bias evaluation + fix_high + sumcheck)
    assert int(r.stdout.split("eqsum_tail: ")[1].split()[0]) >= 2, r.stdout  # Dev::eqsum_tail (eq tables + accumulation sumcheck)
    assert int(r.stdout.split("commit_tail: ")[1].split()[0]) >= 1, r.stdout  # Dev::commit_tail (the last rounds of the Basefold commit phase)
    assert "verify(oracle): ACCEPT" in r.stdout and "verify(product): ACCEPT" in r.stdout


@pytest.mark.parametrize("offset", [3, -1000, 777, -20000])
def test_tampered_proof_rejected(hostlogic_bin, offset):
    r = run(hostlogic_bin, 16, 2, offset)
    assert "verify(oracle,tampered): REJECT" in r.stdout, r.stdout
    assert "verify(product): ACCEPT" in r.stdout


@pytest.mark.parametrize("seed", [1, 4])
def test_cnn_proof_stream_identical_to_oracle_and_accepted(hostlogic_bin, seed):
    """conv (FFT protocol) -> requant -> relu -> maxpool -> flatten -> dense -> requant"""
    r = run(hostlogic_bin, "cnn", seed)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "identical=1" in r.stdout
    assert "verify(oracle): ACCEPT" in r.stdout and "verify(product): ACCEPT" in r.stdout


@pytest.mark.parametrize("where", ["@40", "@900", "@5000", "-3000"])
def test_cnn_tampered_proof_rejected(hostlogic_bin, where):
    r = run(hostlogic_bin, "cnn", 2, where)
    assert "verify(oracle,tampered): REJECT" in r.stdout, r.stdout
    assert "verify(product): ACCEPT" in r.stdout


def test_host_fibers(tmp_path):
    """deep-prove_amd/csrc/fiber.h: the cooperative fibers that let one host thread drive several proofs in flight"""
    import os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = str(tmp_path / "fiber_check")
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-pthread", "-o", exe, os.path.join(root, "tests", "support", "fiber_check.cpp")])
    r = run(exe)
    assert r.returncode == 0 and "fibers ok=1" in r.stdout, r.stdout + r.stderr


def test_field_header_against_wide_arithmetic(tmp_path):
    """deep-prove_amd/csrc/gl64.h (shared by the host orchestrator and every kernel): branch-free add / sub / 128-bit
    reduction / extension product against plain `%` arithmetic on edge values and a random sweep"""
    import os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = str(tmp_path / "field_check")
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-o", exe, os.path.join(root, "tests", "support", "field_check.cpp")])
    r = run(exe)
    assert r.returncode == 0 and "bad=0" in r.stdout, r.stdout + r.stderr


@pytest.mark.parametrize("nv,seed,fs", [(3, 11, "0"), (6, 12, "0"), (9, 13, "0"), (6, 14, "1"), (11, 15, "1")])
def test_generalised_sumcheck_seam_matches_oracle(hostlogic_bin, nv, seed, fs):
    """what dp_sumcheck_prove accepts since round 2: products of 1..5 tables (sumcheck/src/prover.rs:706-713), tables with fewer
    variables than the polynomial (sumcheck_macro/src/lib.rs:236-247: the 2^missing factor, constants once folded out),
    base and extension tables mixed — messages, final evaluations and the transcript after the proof equal the oracle's"""
    r = run(hostlogic_bin, "sumcheck", seed, nv, env={"DP_DOUBLE_DEVICE_FS": fs})
    assert r.returncode == 0, r.stdout + r.stderr
    assert "messages identical=1 finals identical=1 transcript identical=1 verifier=ACCEPT" in r.stdout


@pytest.mark.parametrize("seed,nv,world", [(1, 6, 1), (2, 6, 2), (3, 8, 4), (4, 9, 8), (5, 4, 8)])
def test_in_library_sharded_sumcheck_matches_unsharded_oracle(hostlogic_bin, seed, nv, world):
    """csrc/sharded.h — the per-rank round loop of the sharded prover (what dp_sumcheck_prove_sharded runs over RCCL) — with W
    ranks as W threads over W CPU doubles and an in-memory exchange: every rank's messages, final evaluations and transcript state
    equal the oracle's UNSHARDED prove_parallel of the whole tables (base and extension tables, three terms, stage 2 included)"""
    r = run(hostlogic_bin, "sharded", seed, nv, world)
    assert r.returncode == 0, r.stdout + r.stderr
    assert f"{world} of {world} ranks identical" in r.stdout


@pytest.mark.parametrize("seed,nv,ext", [(1, 8, 0), (2, 9, 1), (3, 10, 0), (4, 12, 1), (5, 13, 0)])
def test_single_polynomial_open_matches_oracle_and_verifies(hostlogic_bin, seed, nv, ext):
    """PCS::open / PCS::verify of ONE polynomial above the trivial size (mpcs/src/basefold.rs:466-544, 863-962; the reference's
    commit_open_verify round trip, mpcs/src/lib.rs:467-540, base and extension): the product's pcs_open over the CPU double gives
    the oracle's stream (root, messages, roots, final message, 200 queries with paths) and leaves the transcript in the oracle's
    state; the product's pcs_verify accepts it with its transcript in the same state, and rejects a wrong evaluation, a foreign
    root, other PCS parameters and four flipped words"""
    r = run(hostlogic_bin, "open", seed, nv, ext)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "identical to the oracle" in r.stdout and "accepted 1 of 1, rejected 7 of 7" in r.stdout


@pytest.mark.parametrize("seed", [1, 2, 9])
def test_matmul_model_proof_stream_identical_to_oracle_and_accepted(hostlogic_bin, seed):
    """MatMul with a constant right matrix (layers/matrix_mul.rs:701-873, 1048-1139; the Linear layer of a transformer block applied
    to every row of a [seq][features] activation): three MatMul (+bias / no bias) + Requant + ReLU blocks over an [8][4] input — the
    product's orchestrator over the CPU double (Dev::fix_high on the activation, Dev::fix_low on the weights, the degree-2 sumcheck,
    the claims routed to the previous layer / the weight and bias commitments) gives the oracle's stream; the verifier accepts both"""
    # the last MatMul with Config::TransposeB (constant matrix stored [n][k]: fix_high, other claim point); an Add with a static operand
    # (layers/add.rs:81-145, 586-625: the learned positional table of transformer/positional.rs) in front of the first MatMul
    # ... and Embeddings as the first layer (layers/transformer/embeddings.rs:359-462, 473-571: tokens in, the one-hot claim checked
    # by the verifier against the public tokens)
    for env in ({}, {"HL_TRANSPOSE": "1"}, {"HL_POSITIONAL": "1"}, {"HL_POSITIONAL": "1", "HL_TRANSPOSE": "1"}, {"HL_EMBED": "1"}, {"HL_EMBED": "1", "HL_POSITIONAL": "1"},
                # Positional::Learned (transformer/positional.rs:327-452, 480-583): a table 1x / 4x as long as the sequence; the slice claim lifted to the table
                {"HL_LEARNED_POS": "1"}, {"HL_LEARNED_POS": "4"}, {"HL_EMBED": "1", "HL_LEARNED_POS": "8"}):
        r = run(hostlogic_bin, "seq", seed, env=env)
        assert r.returncode == 0, r.stdout + r.stderr
        assert "identical=1" in r.stdout
        assert "verify(oracle): ACCEPT" in r.stdout and "verify(product): ACCEPT" in r.stdout


@pytest.mark.parametrize("where", ["@40", "@200", "@2000", "5"])
def test_matmul_model_tampered_proof_rejected(hostlogic_bin, where):
    r = run(hostlogic_bin, "seq", 3, where)
    assert "verify(oracle,tampered): REJECT" in r.stdout, r.stdout + r.stderr


@pytest.mark.parametrize("variant", [0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10])
@pytest.mark.parametrize("seed", [1, 7])
def test_graph_model_proof_stream_identical_to_oracle_and_accepted(hostlogic_bin, variant, seed):
    """Models that are GRAPHS (layers/provable/mod.rs:195-565; Prover::prove over the backward node iterator, iop/prover.rs:437-461): 0 / 1 =
    MatMul of two input tensors (plain / TransposeB, layers/matrix_mul.rs:633-873) -> Add with a third input (layers/add.rs:81-145) -> Requant
    -> ReLU; 2 = QKV (layers/transformer/qkv.rs:462-630) with TWO model outputs, Q and K + 2 V; 3 / 4 = QKV -> ConcatMatMul (Q_h K_h^T per head,
    layers/concat_matmul.rs:467-566, inputs re-laid by their (concat, mat_mul, output) axes) -> ConcatMatMul (scores_h V_h, output permuted
    back to [s][h][d]; 4: the scores stored transposed) -> Add with a second input. The product's orchestrator over the CPU double gives the
    oracle's stream; the verifier — fed from the serialised verifier context, graph section included — accepts both. 5 / 6: LayerNorm (N = 16; N =
    12 of a padded 16) -> shift-only Requant -> ReLU (layers/transformer/layernorm.rs); 7 / 8: Softmax over [heads][n][n] scores under the causal
    mask (layers/transformer/softmax.rs), without and with a zero table. 9 / 10: the reference's Mha layer as ONE node with three inputs
    (layers/transformer/mha.rs:633-724: final_mul, softmax, qk under one node id, one MhaProof) behind a QKV, two heads of 8 / one head of 16."""
    r = run(hostlogic_bin, "graph", variant, seed)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "identical=1" in r.stdout
    assert "verify(oracle): ACCEPT" in r.stdout and "verify(product): ACCEPT" in r.stdout


@pytest.mark.parametrize("variant", [0, 2, 3])
def test_graph_model_layer_proofs_reject_every_flipped_word(hostlogic_bin, variant):
    """a single flipped bit in any of the first words of the stream (the layer proofs: sumcheck messages, final evaluations, the QKV
    aggregation, the pre-bias evaluations) makes the verifier refuse"""
    for at in list(range(1, 60)) + list(range(60, 330, 9)):
        r = run(hostlogic_bin, "graph", variant, 11, f"@{at}")
        assert "verify(oracle,tampered): REJECT" in r.stdout, (at, r.stdout + r.stderr)


# DP_FULL_SWEEPS=1: both variants of every layer and the dense strides (the default keeps the CPU suite short: one variant, every 2nd flip position)
_FULL = bool(int(__import__("os").environ.get("DP_FULL_SWEEPS", "0")))


@pytest.mark.parametrize("variant", [5, 6] if _FULL else [6])
def test_layernorm_proofs_reject_every_flipped_word(hostlogic_bin, variant):
    """LayerNorm (layers/transformer/layernorm.rs:729-1100 / 1230-1505; variants 5 / 6 of the graph models: N = 16, N = 12 of a padded 16) ->
    shift-only Requant -> ReLU: one proof, a single-bit flip in every 5th of the first 6000 words (the LayerNorm proof: two lookups,
    commitments, the accumulation / io / input sumchecks, their evaluations; then Requant and ReLU) and in a sample of the rest (table proofs
    with the committed inverse-square-root column, openings) — the verifier refuses each"""
    import os, re, subprocess
    for sweep in (("1:6000:5", "6000:140000:997") if _FULL else ("1:6000:11", "6000:140000:1999")):
        r = subprocess.run([hostlogic_bin, "graph", str(variant), "11"], capture_output=True, text=True, timeout=900, env=dict(os.environ, DP_FLIP_SWEEP=sweep))
        assert r.returncode == 0 and "identical=1" in r.stdout, r.stdout + r.stderr
        m = re.search(r"flip sweep: (\d+) flipped, (\d+) rejected, accepted at:(.*)", r.stdout)
        assert m and int(m.group(1)) > (100 if _FULL else 30) and m.group(1) == m.group(2) and not m.group(3).strip(), r.stdout


@pytest.mark.parametrize("variant", [7, 8] if _FULL else [8])
def test_softmax_proofs_reject_every_flipped_word(hostlogic_bin, variant):
    """Softmax (layers/transformer/softmax.rs:573-888 / 1274-1586; variants 7 / 8: without / with a zero table): one proof, a single-bit flip in
    every 3rd of the first 6000 words (the four lookups, the commitments, the accumulation and the mask sumcheck, the evaluations) and in a
    sample of the rest (table proofs with the committed exponential / error columns, openings) — the verifier refuses each"""
    import os, re, subprocess
    for sweep in (("1:6000:3", "6000:74000:211") if _FULL else ("1:6000:7", "6000:74000:431")):
        r = subprocess.run([hostlogic_bin, "graph", str(variant), "11"], capture_output=True, text=True, timeout=900, env=dict(os.environ, DP_FLIP_SWEEP=sweep))
        assert r.returncode == 0 and "identical=1" in r.stdout, r.stdout + r.stderr
        m = re.search(r"flip sweep: (\d+) flipped, (\d+) rejected, accepted at:(.*)", r.stdout)
        assert m and int(m.group(1)) > (100 if _FULL else 30) and m.group(1) == m.group(2) and not m.group(3).strip(), r.stdout


@pytest.mark.parametrize("variant", [9, 10] if _FULL else [9])
def test_mha_proofs_reject_every_flipped_word(hostlogic_bin, variant):
    """Mha as one node (layers/transformer/mha.rs:633-724 / 792-893; variants 9 / 10): one proof, a single-bit flip in every 5th of the first
    8000 words (the QKV proof, then the MhaProof: final_mul's sumcheck and claims, the softmax's four lookups, commitments, accumulation and
    mask sumchecks, evaluations, qk's sumcheck and claims) and in a sample of the rest — the verifier refuses each"""
    import os, re, subprocess
    for sweep in (("1:8000:3", "8000:106000:307") if _FULL else ("1:8000:11", "8000:106000:997")):
        r = subprocess.run([hostlogic_bin, "graph", str(variant), "11"], capture_output=True, text=True, timeout=900, env=dict(os.environ, DP_FLIP_SWEEP=sweep))
        assert r.returncode == 0 and "identical=1" in r.stdout, r.stdout + r.stderr
        m = re.search(r"flip sweep: (\d+) flipped, (\d+) rejected, accepted at:(.*)", r.stdout)
        assert m and int(m.group(1)) > (100 if _FULL else 30) and m.group(1) == m.group(2) and not m.group(3).strip(), r.stdout


def test_batch_commit_and_simple_batch_open_over_the_double(hostlogic_bin):
    """PCS::batch_commit + simple_batch_open (mpcs/src/basefold.rs:356-446, 777-861): the product's host code over the CPU double —
    encode every polynomial, the common tree as the ordinary tree over the row hashes (Dev::batch_tree), the commit phase on the
    eq(t)-weighted sums, the row-pair queries with one path — gives the oracle's root, stream and transcript state for base and extension
    polynomials, batches of 1..9, trivial sizes; pcs_simple_batch_verify accepts, rejects wrong evaluations / order / root / field /
    parameters and every one of ~100 single-word flips spread over the stream"""
    for args in ((1, 9, 0, 3), (2, 9, 1, 3), (3, 10, 0, 5), (4, 9, 1, 2), (5, 8, 0, 1), (6, 8, 0, 4), (7, 5, 0, 4), (8, 5, 1, 3), (9, 3, 0, 7), (10, 1, 0, 2), (11, 11, 0, 9), (12, 9, 1, 1)):
        r = subprocess.run([hostlogic_bin, "batchopen", *map(str, args)], capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stdout + r.stderr
        assert "identical to the oracle" in r.stdout and "accepted 1 of 1, rejected 5 of 5" in r.stdout, r.stdout


def test_batch_open_over_general_evaluation_lists(hostlogic_bin):
    """PCS::batch_open with `evals: &[Evaluation]` (mpcs/src/basefold.rs:546-770; the shapes of run_batch_commit_open_verify{,_multiple_sizes},
    mpcs/src/lib.rs:508-700, a polynomial at two points, mixed lists): the product runs the classic sumcheck on one (f, eq) pair per evaluation
    and the commit / query phases per commitment; the oracle merges the polynomials per point as the reference does — same stream, same
    transcript; pcs_batch_verify_evals accepts, rejects a wrong or missing evaluation and every sampled single-word flip"""
    for shape in range(5):
        r = subprocess.run([hostlogic_bin, "batchevals", str(shape + 3), str(shape)], capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stdout + r.stderr
        assert "identical to the oracle" in r.stdout and "accepted 1 of 1, rejected 2 of 2" in r.stdout, r.stdout


def _graph_golden():
    import json, os
    with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "graph_models.json")) as f:
        return json.load(f)


@pytest.mark.parametrize("case", [0, 2, 4, 6, 8, 11, 13, 15, 16, 17])  # (a model of every kind; the others run on the GPU, tests/test_gpu_model.py)
def test_golden_graph_blobs_through_the_products_blob_parser_on_the_cpu_double(hostlogic_bin, tmp_path, case):
    """every model of tests/golden/graph_models.json as dp_model_setup receives it: the int64 blob models.py writes, read by the product's own
    parser (csrc/blob.h, the one behind the C ABI) and proved by the product's orchestrator over the CPU double, gives the oracle's stream (whose
    sha256 the golden pins, tests/test_oracle.py) word for word; the verifier, fed from the serialised verifier context, accepts it and refuses a
    flipped word. Cases 11 / 12: the Mha node; 13: a whole transformer layer (19 nodes); 15 - 17: Activation::Gelu (15: the reference's own test
    shape, the oracle to the letter of the reference's prover; 16 / 17: with the claim the reference's VERIFIER checks, see test_gelu_* below)."""
    import subprocess
    import numpy as np
    import deep_prove_amd as dpa
    c = _graph_golden()[case]
    g = getattr(dpa.models, c["model"])(**c["args"])
    bp, ip = tmp_path / "model.blob", tmp_path / "input.bin"
    g.blob().astype(np.int64).tofile(bp)
    g.input().astype(np.int64).tofile(ip)
    env = dict(os.environ, HL_GELU_LITERAL="1") if c.get("oracle_gelu_claim") == "reference" else dict(os.environ)
    r = subprocess.run([hostlogic_bin, "blob", str(bp), str(ip)], capture_output=True, text=True, timeout=900, env=env)
    assert r.returncode == 0, r.stdout + r.stderr
    assert f"oracle words={c['proof_words']} product words={c['proof_words']} identical=1" in r.stdout, r.stdout
    assert "verify(oracle): ACCEPT" in r.stdout and "verify(product): ACCEPT" in r.stdout
    if case >= 11:
        r = subprocess.run([hostlogic_bin, "blob", str(bp), str(ip), "@77"], capture_output=True, text=True, timeout=900)
        assert "verify(oracle,tampered): REJECT" in r.stdout, r.stdout + r.stderr


@pytest.mark.parametrize("variant", [11, 12, 13])
def test_gelu_proof_stream_identical_to_oracle_and_accepted(hostlogic_bin, variant):
    """Activation::Gelu (zkml/src/layers/activation.rs:238-318 witness, :385-456 prove_step, :459-517 verify_activation; the table of
    lookup/context.rs:163-182, 364-378, 495-503): 11 = the reference's own proving test (one GELU over a small tensor), 12 = 256 entries, 13 = Dense ->
    Requant -> GELU (multiplier 45, a table of 2^14 rows) -> Dense -> Requant. Product stream over the CPU double = the oracle's, both accepted."""
    r = run(hostlogic_bin, "graph", variant, 5)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "identical=1" in r.stdout and "verify(oracle): ACCEPT" in r.stdout and "verify(product): ACCEPT" in r.stdout


def test_gelu_to_the_letter_of_the_reference_only_verifies_when_the_column_is_shown(hostlogic_bin):
    """The reference's GELU prover hands its commitment prover the lookup claim DIVIDED by the multiplier (activation.rs:405-430: `input_claim` is re-bound
    before the `commits` array is built) where its verifier files the lookup's own claim (:495-505). With the oracle to the letter (HL_GELU_LITERAL=1):
    variant 11 — columns of 2^5 entries, opened by showing them (Basefold::open ignores the evaluation, mpcs/src/basefold.rs:466-483) — is the product's
    stream word for word and verifies; variant 12 — 2^8 entries, a batch opening that starts from the claimed evaluations (basefold.rs:601-685) — differs
    from the product's stream and the verifier refuses it, while the product's own proof of the same model is accepted."""
    env = dict(os.environ, HL_GELU_LITERAL="1")
    r = subprocess.run([hostlogic_bin, "graph", "11", "5"], capture_output=True, text=True, timeout=900, env=env)
    assert r.returncode == 0 and "identical=1" in r.stdout and "verify(oracle): ACCEPT" in r.stdout and "verify(product): ACCEPT" in r.stdout, r.stdout + r.stderr
    r = subprocess.run([hostlogic_bin, "graph", "12", "5"], capture_output=True, text=True, timeout=900, env=env)
    assert "identical=0" in r.stdout and "verify(oracle): REJECT" in r.stdout and "verify(product): ACCEPT" in r.stdout, r.stdout + r.stderr


def test_gelu_reference_letter_switch(hostlogic_bin):
    """DP_GELU_REFERENCE_LETTER=1 (csrc/zkml.h prove_relu): the PRODUCT files the descaled claim with the scaled column's commitment, as the reference's prover does
    (activation.rs:419-430). Against the oracle to the letter (HL_GELU_LITERAL=1): variant 11 (2^5 entries, shown) and 13 (Dense -> Requant -> GELU -> ..., a table of 2^14 rows, a column of 2^6 entries)
    identical and accepted; variant 12 (2^8 entries) identical word for word — the reference prover's stream — and refused by the verifier, like the reference's own
    proof would be."""
    env = dict(os.environ, HL_GELU_LITERAL="1", DP_GELU_REFERENCE_LETTER="1")
    r = subprocess.run([hostlogic_bin, "graph", "11", "5"], capture_output=True, text=True, timeout=900, env=env)
    assert r.returncode == 0 and "identical=1" in r.stdout and "verify(oracle): ACCEPT" in r.stdout and "verify(product): ACCEPT" in r.stdout, r.stdout + r.stderr
    r = subprocess.run([hostlogic_bin, "graph", "12", "5"], capture_output=True, text=True, timeout=900, env=env)
    assert "identical=1" in r.stdout and "verify(oracle): REJECT" in r.stdout and "verify(product): REJECT" in r.stdout, r.stdout + r.stderr
    r = subprocess.run([hostlogic_bin, "graph", "13", "5"], capture_output=True, text=True, timeout=900, env=env)  # (its GELU column has 2^6 entries: shown, so the claim is never used)
    assert "identical=1" in r.stdout and "verify(oracle): ACCEPT" in r.stdout and "verify(product): ACCEPT" in r.stdout, r.stdout + r.stderr


def test_gelu_layer_proofs_reject_every_flipped_word(hostlogic_bin):
    """every third word of the first 6000 of a GELU proof (variant 13: the activation's lookup, its accumulation sumcheck, the table proof with the
    committed output column) flipped, each through the full verifier: all refused"""
    r = subprocess.run([hostlogic_bin, "graph", "13", "11"], capture_output=True, text=True, timeout=900, env=dict(os.environ, DP_FLIP_SWEEP="2:6000:3"))
    assert "accepted at:\n" in r.stdout or r.stdout.rstrip().endswith("accepted at:"), r.stdout[-600:]
    assert "flip sweep:" in r.stdout


def test_malformed_model_blobs_are_refused_by_the_products_parser(hostlogic_bin, tmp_path):
    """the blob parser behind dp_model_setup on truncated / mutated blobs of the Mha block: refused with an error (or, when the mutation leaves a
    well-formed model, proved) — never a crash"""
    import subprocess
    import numpy as np
    import deep_prove_amd as dpa
    g = dpa.models.mha_block(8, 16, 2, 8, config=97)
    blob, x = g.blob().astype(np.int64), g.input().astype(np.int64)
    ip = tmp_path / "input.bin"
    x.tofile(ip)
    rng = np.random.default_rng(5)
    mha_at = [i for i in range(blob.size - 4) if blob[i] == 16 and blob[i + 1] == 3][0]  # [16, three inputs, ...]
    variants = [blob[:mha_at + 5], blob[:-3], np.concatenate([blob, [1, 2]])]
    for off, val in ((1, 4), (1, 0), (8, 3), (9, 5), (10, 0), (14, 40), (16, 7)):  # input count, shape, table size, zero chunks ...
        b = blob.copy(); b[mha_at + off] = val; variants.append(b)
    for _ in range(2):
        b = blob.copy(); b[int(rng.integers(0, 60))] = int(rng.integers(-3, 70)); variants.append(b)
    refused = 0
    for k, b in enumerate(variants):
        bp = tmp_path / f"m{k}.blob"
        b.tofile(bp)
        r = subprocess.run([hostlogic_bin, "blob", str(bp), str(ip)], capture_output=True, text=True, timeout=900)
        assert r.returncode in (0, 1, 2, 4), (k, r.returncode, r.stdout[-300:] + r.stderr[-300:])  # (no signal: negative return codes)
        refused += r.returncode == 4 or "refused" in r.stdout or r.returncode == 1
    assert refused >= 6, refused
