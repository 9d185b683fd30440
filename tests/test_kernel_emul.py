"""Device SOURCE of deep-prove_amd/csrc/hip_dev.hip executed on the CPU (tests/support/kernel_emul: a fiber-based SIMT
emulator with __syncthreads, wave shuffles and the DPP moves of the lane-parallel Poseidon2). What it pins without a GPU:
k_logup_tail — the whole logup-GKR layer loop (or, in full mode, the whole logup-GKR proof) with device-side Fiat-Shamir in one
launch (Dev::logup_tail / Dev::logup_full) —, k_classic_tail (the last rounds of the batch-opening sumcheck) k_dense_tail (bias
evaluation + fix_high + sumcheck of a Dense layer) k_eqsum_tail (eq tables + accumulation sumcheck) and k_commit_tail (FRI fold + merges + sumcheck pairs +
Merkle trees + roots of the last Basefold commit rounds), together with
everything it is built from (sc_accumulate, sc_fs_round, the wave sponge wc_*, p2l_permute, wg_build_eq) and the product's
host code on both sides of the launch (csrc/logup_tail.h), byte for byte against the layer-by-layer path."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_logup_tail_kernel_on_the_simt_emulator():
    sys.path.insert(0, ROOT)
    import __graft_entry__ as g
    binary = g.build_kernel_emul()
    r = subprocess.run([binary], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout + r.stderr
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith(("tail n=", "full n="))]
    assert len(lines) >= 30 and sum(ln.startswith("full") for ln in lines) >= 10  # hand-picked + 16 seeded random shapes
    for ln in lines:
        assert "kernel taken=1 declined=0" in ln and "identical=1" in ln and "transcript_after=1" in ln, ln
    assert "identical=0" not in r.stdout
    assert any("table" in ln for ln in lines) and any("threads=1024" in ln for ln in lines) and any("threads=256" in ln for ln in lines)


def _model(args, env):
    sys.path.insert(0, ROOT)
    import __graft_entry__ as g
    g.build_kernel_emul()
    binary = os.path.join(ROOT, "tests", "support", "kernel_emul", "_build", "hostlogic_check_emul")
    e = dict(os.environ)
    e.update(env)
    return subprocess.run([binary, *map(str, args)], capture_output=True, text=True, timeout=900, env=e)


def test_factored_eq_rounds_of_the_batch_opening_down_to_two_entries():
    """k_classic_fused on FACTORED eq tables (classic_fused_factored: eq(x, z) as the outer product of two short tables, Dev::classic_round)
    through every shape of the pair — the low factor folding (4 and more entries), becoming a scalar (2), the high factor folding under
    the scalar (1), the last fold to one entry — with no classic tail taking over: the proofs equal the oracle's byte for byte"""
    for args in ((16, 3), ("cnn", 2)):
        r = _model(args, {"DP_EMUL_CLASSIC": "0"})
        assert r.returncode == 0, r.stdout + r.stderr
        assert "identical=1" in r.stdout, r.stdout
        assert int(r.stdout.split("emulated k_classic_tail: ")[1].split()[0]) == 0, r.stdout
        fused = r.stdout.split("emulated k_classic_fused: ")[1].split()
        assert int(fused[0]) >= 8 and int(fused[2]) >= 8, r.stdout
        assert "verify(oracle): ACCEPT" in r.stdout and "verify(product): ACCEPT" in r.stdout


def test_whole_model_proofs_with_every_logup_proof_from_the_emulated_kernel():
    """end to end: the product's orchestrator proves an MLP and a CNN with EVERY logup-GKR proof (lookups and tables) produced
    by the device source of k_logup_tail on the emulator — full mode (one launch per proof) and tail mode — and the proof
    streams equal the oracle's byte for byte; the verifier accepts them"""
    for args, env, least in (((64, 1), {"DP_EMUL_COMMIT_MAX_N": "4096"}, 13), ((16, 7), {"DP_EMUL_MODE": "1", "DP_EMUL_THREADS": "256"}, 13), (("cnn", 4), {"DP_EMUL_THREADS": "256"}, 10)):
        r = _model(args, env)
        assert r.returncode == 0, r.stdout + r.stderr
        assert "identical=1" in r.stdout, r.stdout
        taken = int(r.stdout.split("logup proofs taken")[0].split()[-1])
        declined = int(r.stdout.split("logup proofs taken, ")[1].split()[0])
        assert taken >= least and declined == 0, r.stdout
        assert int(r.stdout.split("emulated k_classic_tail: ")[1].split()[0]) >= 1, r.stdout  # ... and the batch-opening sumcheck tail
        fused = r.stdout.split("emulated k_classic_fused: ")[1].split()  # ... and the streamed rounds before it, on factored eq tables
        assert int(fused[0]) >= 1 and int(fused[2]) >= 1, r.stdout
        assert int(r.stdout.split("k_eq_outer_many: ")[1].split()[0]) >= 1, r.stdout          # ... written out (k_eq_outer_many) where the tail takes over
        ax = r.stdout.split("emulated k_axpy_many: ")[1].split()  # ... and the batch opening's sums over codewords / evaluation tables, the short ones class by class first (k_axpy_classes)
        assert int(ax[0]) >= 2 and int(ax[2]) >= 1 and int(ax[7]) >= 2, r.stdout
        assert int(r.stdout.split("emulated k_dense_tail: ")[1].split()[0]) >= 1, r.stdout    # ... and every Dense layer (bias, fix_high, sumcheck)
        assert int(r.stdout.split("emulated k_eqsum_tail: ")[1].split()[0]) >= 2, r.stdout    # ... and the accumulation sumchecks of Requant / ReLU
        if args[0] == "cnn":  # ... and the delegation chains of the FFT / FFT-of-weights / iFFT of the convolution, each in one launch (k_deleg_tail)
            dg = r.stdout.split("emulated k_deleg_tail: ")[1].split()
            assert int(dg[0]) == 3 and int(dg[4].strip("(")) == 24, r.stdout
        assert int(r.stdout.split("emulated k_commit_tail: ")[1].split()[0]) >= 1, r.stdout   # ... and the last rounds of the Basefold commit phase
        if "DP_EMUL_COMMIT_MAX_N" in env:  # several rounds in one launch: FRI folds, messages and Merkle trees, not only the final round
            assert int(r.stdout.split("commit-phase tails taken (")[1].split()[0]) >= 4, r.stdout
        assert "verify(oracle): ACCEPT" in r.stdout and "verify(product): ACCEPT" in r.stdout


def test_gelu_models_and_the_512_thread_form_on_the_emulated_kernels():
    """Activation::Gelu through the DEVICE source on the emulator: the lookup over (input * multiplier, output) and the table proof of a 2^13- / 2^14-row GELU
    table with its committed output column come out of the emulated k_logup_tail (graph variants 11 and 13 of hostlogic_check), streams equal to the oracle's,
    verifier accepts; and the same kernels at 512 threads — the thread count of the throughput-mode form for lookups of >= 2 048 rows (DPL_ONE_W, hip_dev.hip)"""
    for args, env, least in ((("graph", 11, 5), {"DP_EMUL_THREADS": "256"}, 2), (("graph", 13, 5), {"DP_EMUL_THREADS": "512"}, 9)):
        r = _model(args, env)
        assert r.returncode == 0, r.stdout + r.stderr
        assert "identical=1" in r.stdout and f"{env['DP_EMUL_THREADS']} threads" in r.stdout, r.stdout
        taken = int(r.stdout.split("logup proofs taken")[0].split()[-1])
        declined = int(r.stdout.split("logup proofs taken, ")[1].split()[0])
        assert taken >= least and declined == 0, r.stdout
        assert "verify(oracle): ACCEPT" in r.stdout and "verify(product): ACCEPT" in r.stdout


def test_batch_opening_sums_with_and_without_class_sums():
    """k_axpy_many / k_axpy_classes (csrc/axpy_many.h plans them): the polynomials shorter than the accumulator summed among their own length first, or — DP_EMUL_AXPY_CLASSES=0,
    the product's DP_AXPY_CLASSES=0 — every descriptor in the one pass; three workgroups of 64 lanes, so the four-elements-per-lane body and the remainder loop both run.
    Either way the proof equals the oracle's byte for byte"""
    for env, grouped in (({}, True), ({"DP_EMUL_AXPY_CLASSES": "0"}, False)):
        r = _model((16, 3), env)
        assert r.returncode == 0, r.stdout + r.stderr
        assert "identical=1" in r.stdout and "verify(product): ACCEPT" in r.stdout, r.stdout
        ax = r.stdout.split("emulated k_axpy_many: ")[1].split()
        assert int(ax[0]) >= 3 and (int(ax[2]) >= 1) == grouped, r.stdout


def test_single_polynomial_open_with_the_emulated_commit_tail():
    """PCS::open of one polynomial (mpcs/src/basefold.rs:466-544) with the last commit rounds served by the device source of
    k_commit_tail on the emulator (only the final round, and three / four rounds in one launch): stream, root and transcript
    state equal the oracle's single-polynomial commit phase + query phase"""
    for args, env, rounds in ((("open", 3, 9, 0), {}, 1), (("open", 5, 11, 0), {"DP_EMUL_COMMIT_MAX_N": "4096"}, 3), (("open", 7, 12, 1), {"DP_EMUL_COMMIT_MAX_N": "4096", "DP_EMUL_THREADS": "256"}, 4)):
        r = _model(args, env)
        assert r.returncode == 0, r.stdout + r.stderr
        assert "identical to the oracle" in r.stdout and "accepted 1 of 1, rejected 7 of 7" in r.stdout
        assert f"1 commit-phase tails taken ({rounds} rounds)" in r.stdout, r.stdout


def test_fused_kernels_with_the_sponge_on_the_host():
    """DP_HOST_SPONGE (csrc/sponge_host.h): the five fused protocol kernels run their WaveChallenger in host mode — observed words
    staged in a request area, challenges fetched from a reply area, the requests served by the product's own service
    (sponge_serve_all, here called from inside the kernel's reply poll) on the host transcript — and the proofs of an MLP (single- and
    multi-round commit tails) and of the CNN still equal the oracle's streams byte for byte"""
    for args, env in (((16, 7), {}), ((64, 1), {"DP_EMUL_COMMIT_MAX_N": "4096"}), (("cnn", 4), {"DP_EMUL_THREADS": "256"}), (("seq", 2), {})):
        e = {"DP_EMUL_HOST_SPONGE": "1"}
        e.update(env)
        r = _model(args, e)
        assert r.returncode == 0, r.stdout + r.stderr
        assert "identical=1" in r.stdout and "verify(product): ACCEPT" in r.stdout, r.stdout
        ran = int(r.stdout.split("host sponge service: ")[1].split()[0])
        served = int(r.stdout.split("sponge on the host, ")[1].split()[0])
        assert ran >= 10 and served > 10 * ran, r.stdout


def test_batch_tree_row_hashes_from_the_emulated_kernel():
    """dp_pcs_batch_commit's only own kernel, k_batch_row_hash (hash_or_noop of every row of the batch tree: pass-through up to four words,
    the overwrite-mode sponge above), from the device source on the emulator — three blocks of 64 lanes, grid-stride — inside the product's
    batch_commit + simple_batch_open: root, stream and transcript equal the oracle's; base / extension, 2..9 polynomials, trivial sizes"""
    for args in ((1, 9, 0, 3), (2, 9, 1, 3), (3, 10, 0, 5), (4, 9, 1, 2), (6, 8, 0, 4), (7, 5, 0, 4), (8, 5, 1, 3), (9, 3, 0, 7), (11, 10, 0, 9), (13, 8, 1, 5)):
        r = _model(("batchopen",) + args, {})
        assert r.returncode == 0, r.stdout + r.stderr
        assert "identical to the oracle" in r.stdout and "accepted 1 of 1, rejected 5 of 5" in r.stdout, r.stdout
        assert f"emulated k_batch_row_hash: {2 << args[1] if args[1] > 7 else 1 << args[1]} rows hashed" in r.stdout, r.stdout


def test_general_evaluation_lists_with_the_emulated_sumcheck_tail():
    """PCS::batch_open over general Evaluation lists with the classic-sumcheck tail and the commit tail from the device source on the emulator"""
    for shape in (1, 3):
        r = _model(("batchevals", shape + 3, shape), {})
        assert r.returncode == 0, r.stdout + r.stderr
        assert "identical to the oracle" in r.stdout and "accepted 1 of 1, rejected 2 of 2" in r.stdout, r.stdout


def test_logup_tail_threshold_knob_moves_long_lookups_to_the_layer_kernels_without_changing_the_proof():
    """DP_LOGUP_TAIL_MAX_N (csrc/logup_tail.h: the longest lookup column k_logup_tail proves in one workgroup): with 64, the MLP's longer lookups
    are declined by the tail and run layer by layer, the shorter ones still come from the emulated kernel — and the proof stream is the oracle's"""
    r = _model((64, 1), {"DP_LOGUP_TAIL_MAX_N": "64"})
    assert r.returncode == 0, r.stdout + r.stderr
    assert "identical=1" in r.stdout and "verify(product): ACCEPT" in r.stdout, r.stdout
    taken = int(r.stdout.split("logup proofs taken")[0].split()[-1])
    declined = int(r.stdout.split("logup proofs taken, ")[1].split()[0])
    assert taken >= 1 and declined >= 1, r.stdout
