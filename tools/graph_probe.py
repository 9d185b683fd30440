"""a graph model at a larger shape on the device — models.attention_block (QKV, ConcatMatMul x2, MatMul, Add of two inputs, five Requants) or, with
GRAPH_MODEL=transformer_layer / mha_block / transformer_block, the blocks with LayerNorm, Softmax and the Mha node — parity with the oracle for one
proof, then single-proof latency and batch throughput. GRAPH_FFN = width of the feed-forward half of transformer_layer (default 4 x emb)."""
import os, sys, time
_root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, _root); sys.path.insert(0, os.path.join(_root, "tests"))
import numpy as np
import deep_prove_amd as dpa
from support import oracle_lib
seq, emb, heads, hd = (int(a) for a in sys.argv[1:5]) if len(sys.argv) > 4 else (64, 256, 4, 64)
conc = int(sys.argv[5]) if len(sys.argv) > 5 else 128
name = os.environ.get("GRAPH_MODEL", "attention_block")
g = dpa.models.transformer_layer(seq, emb, heads, hd, int(os.environ.get("GRAPH_FFN", 4 * emb)), config=66) if name == "transformer_layer" else getattr(dpa.models, name)(seq, emb, heads, hd, config=66)
dev = dpa.Device(0)
ctx = dpa.Context.generate(dev, g.blob())
pr = dpa.Prover(ctx)
x = g.input()
proof, out = pr.prove(x)
assert (out == g.run(x)).all()
if os.environ.get("GRAPH_NO_ORACLE"):  # (sweeps: parity is established by the run without this switch)
    print(f"{name} ({len(g.nodes)} nodes) seq {seq} emb {emb} heads {heads} x {hd}: proof {proof.size} words (oracle not run), DP_LOGUP_TAIL_MAX_N={os.environ.get('DP_LOGUP_TAIL_MAX_N', 'default')}", flush=True)
else:
    o = oracle_lib.load()
    h = o.model_setup(g.blob())
    t0 = time.perf_counter(); oproof, oout, oms = o.model_prove(h, x); o.model_free(h)
    print(f"{name} ({len(g.nodes)} nodes) seq {seq} emb {emb} heads {heads} x {hd}: proof {proof.size} words, identical to the oracle: {bool(proof.size == oproof.size and (proof == oproof).all())} (oracle {oms:.0f} ms on one core), DP_LOGUP_TAIL_MAX_N={os.environ.get('DP_LOGUP_TAIL_MAX_N', 'default')}", flush=True)
dpa.verify(ctx.verifier_blob(), proof, x, out)
lat = []
for _ in range(3):
    t0 = time.perf_counter(); pr.prove(x); lat.append(1000 * (time.perf_counter() - t0))
xs = np.stack([g.input(100 + i) for i in range(conc * 3)])
pr.prove_batch(xs[:conc], conc)
t0 = time.perf_counter(); proofs, outs, _ = pr.prove_batch(xs, conc); dt = time.perf_counter() - t0
v, _ = dpa.verify_batch(ctx.verifier_blob(), proofs[:16], xs[:16], outs[:16], dev=dev)
print(f"single proof {sorted(lat)[1]:.1f} ms; batch: {len(xs) / dt:.1f} proofs/s ({conc} in flight, {len(xs)} proofs), rejected of 16: {int(v.sum())}, in flight {pr.in_flight()}", flush=True)
