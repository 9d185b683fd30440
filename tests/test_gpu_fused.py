"""Throughput mode at FULL size against the oracle's goldens, and the fused protocol kernels knob by knob.

Throughput mode (dp_model_prove_batch: cohorts, device-side Fiat-Shamir, the one-launch protocol kernels k_logup_tail /
k_classic_tail / k_dense_tail / k_eqsum_tail / k_commit_tail) is what bench.py times. Its proofs must be the proofs the
reference defines, not merely proofs the verifier accepts:
 * the golden input of tests/golden/{dense4m,cnn264k}_proof.json (oracle output, tests/golden/make_*_hash.py) sits at a
   NON-ZERO index of a batch with >= 16 proofs in flight and default knobs, and the sha256 of that proof's canonical stream
   must equal the oracle's;
 * every fused kernel switched off alone, all of them off, and the remaining tuning knobs: proof 0 of a batch equals the
   sequential (latency-mode, host transcript) proof of the same input and sampled proofs verify.
Every knob configuration runs in its own process (the knobs are read when the library creates its contexts)."""
import hashlib
import json
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("workload,gold_name,conc,slot", [("dense_4m", "dense4m_proof.json", 24, 13), ("cnn_264k", "cnn264k_proof.json", 16, 5)])
def test_throughput_mode_proof_is_the_oracles_proof_at_full_size(dev, workload, gold_name, conc, slot):
    import deep_prove_amd as dpa
    gold = json.load(open(os.path.join(ROOT, "tests", "golden", gold_name)))
    mb = getattr(dpa.models, workload)()
    ctx = dpa.Context.generate(dev, mb.blob())
    pr = dpa.Prover(ctx)
    xs = np.stack([mb.input(3000 + i) for i in range(conc + 3)])  # a ragged last cohort as well
    xs[slot] = mb.input(gold["input_index"])
    proofs, outs, _ = pr.prove_batch(xs, conc)
    assert pr.in_flight() == conc
    assert [int(v) for v in outs[slot]] == gold["output"]
    assert proofs[slot].size == gold["proof_words"]
    assert hashlib.sha256(proofs[slot].tobytes()).hexdigest() == gold["sha256"], "throughput-mode proof differs from the oracle's proof stream"
    vb = ctx.verifier_blob()
    for i in (0, slot, len(xs) - 1):
        dpa.verify(vb, proofs[i], xs[i], outs[i])
    ctx.free()


ALL_OFF = {"DP_DEVICE_LOGUP": "0", "DP_FUSED_OFF": "classic,dense,eqsum,commit,deleg"}
KNOBS = [{}, {"DP_DEVICE_LOGUP": "0"}, {"DP_DEVICE_LOGUP": "1"}, {"DP_FUSED_OFF": "classic"}, {"DP_FUSED_OFF": "dense"}, {"DP_FUSED_OFF": "eqsum"},
         {"DP_FUSED_OFF": "commit"}, ALL_OFF,
         {"DP_PREP_THREADS": "0"},  # no helper threads: every worker runs inference and the host half of the witness generation itself when it starts a proof
         {"DP_AXPY_CLASSES": "0"},  # the batch opening's short polynomials straight into the accumulator pass (k_axpy_classes off)
         {"DP_HOST_SPONGE": "1"},  # the fused kernels with the transcript's sponge on the host (csrc/sponge_host.h)
         {"DP_QUERY_SECTION_HOST": "1"},  # the batch opening's query section from host-built descriptors (k_query_gather) instead of k_query_section
         {"DP_LOGUP_WIDE_N": "0"}, {"DP_LOGUP_WIDE_N": "1024"},  # k_logup_tail's 512-thread throughput form: never / from 1 024 rows on (default 2 048)
         {"DP_MAILBOX_VRAM": "0", "DP_LP_MAX": "4096", "DP_NUMA_PIN": "0"}]  # the single proof the batch is compared with: mailbox in host memory, round 4's Merkle threshold, no affinity


def _ident(f):
    return "+".join(f"{k[3:].lower()}={v}" for k, v in f.items()) or "default"


@pytest.mark.parametrize("workload,conc,knobs", [("dense_4m", 16, k) for k in KNOBS] + [("cnn_264k", 8, k) for k in ({}, {"DP_DEVICE_LOGUP": "1"}, ALL_OFF, {"DP_HOST_SPONGE": "1"}, {"DP_FUSED_OFF": "deleg"}, {"DP_LOGUP_WIDE_N": "0"})],  # (deleg off: one launch per delegation sumcheck)
                         ids=lambda v: _ident(v) if isinstance(v, dict) else str(v))
def test_batch_proof_equals_sequential_proof_under_every_knob(workload, conc, knobs):
    env = dict(os.environ)
    env.update(knobs)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "knob_sweep.py"), "--one", workload, str(conc)], capture_output=True, text=True, timeout=300, env=env, cwd=ROOT)
    line = r.stdout.strip().splitlines()[-1] if r.stdout.strip() else ""
    assert r.returncode == 0 and line.startswith("{"), (r.stdout + r.stderr)[-2000:]
    rec = json.loads(line)
    assert rec["batch0_equals_single"] and rec["verified"] == 2, rec
