"""Every lookup TABLE and every lookup WITNESS column of the golden model families, three ways: the product's host code (csrc/zkml.h table_columns /
witness_host), the oracle (oracle/zkml.hpp instantiate_witness_ctx) — both dumped by `hostlogic_check lookups` — and tests/support/lookups_independent.py,
a numpy writing from the reference's text (zkml/src/lookup/context.rs:158-296, 464-503, 631-756 and the layers' gen_lookup_witness). The two C++ copies share
their reading of those files (126 lines verbatim — LUT builders, maxpool / softmax columns, the TableType order): a misreading there passes every
HIP == oracle parity test; it does not pass this one. CPU only."""
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import deep_prove_amd.models as M  # noqa: E402
from support import lookups_independent as LI  # noqa: E402

HARNESS = os.path.join(ROOT, "tests", "support", "_build", "hostlogic_check")


@pytest.fixture(scope="module")
def harness():
    import __graft_entry__ as g
    g.build_oracle()
    g.build_test_harness()
    return HARNESS


def dump(harness, mb, x, tmp_path):
    b, i, o = (str(tmp_path / n) for n in ("blob.bin", "in.bin", "out.bin"))
    np.asarray(mb.blob(), dtype=np.int64).tofile(b)
    np.asarray(x, dtype=np.int64).tofile(i)
    r = subprocess.run([harness, "lookups", b, i, o], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    w = np.fromfile(o, dtype=np.uint64)
    recs, at = {0: [], 1: []}, 0
    while at < w.size:
        tag, src, kind, size, aux, aux2, node, which, nl, nc, rows = (int(v) for v in w[at:at + 11])
        at += 11
        ncols = nl + nc + (1 if tag == 1 else 0)
        cols = w[at:at + ncols * rows].reshape(ncols, rows)
        at += ncols * rows
        recs[src].append(dict(tag=tag, kind=kind, size=size, aux=aux, aux2=np.int64(np.uint64(aux2)).item(), node=node, which=which, nl=nl, nc=nc, cols=cols))
    assert at == w.size
    return recs


def expected(mb, x):
    col = LI.collect(mb, x)
    out = []
    for k, t in enumerate(col.table_records()):
        tt = t["t"]
        out.append(dict(tag=1, kind=tt["kind"], size=tt.get("size", 0), aux=tt.get("aux", 0), aux2=tt.get("aux2", 0), node=0, which=k, nl=len(t["cols"]), nc=0,
                        cols=np.stack([LI.field(c) for c in t["cols"]] + [t["mult"]])))
    for lk in col.lookups:
        tt = lk["t"]
        out.append(dict(tag=2, kind=tt["kind"], size=tt.get("size", 0), aux=tt.get("aux", 0), aux2=tt.get("aux2", 0), node=lk["node"], which=lk["which"],
                        nl=len(lk["lookup"]), nc=len(lk["committed"]), cols=np.stack([LI.field(c) for c in lk["lookup"] + lk["committed"]])))
    return out


def differences(exp, got):
    """human-readable list of what differs between the numpy records and one C++ source's"""
    bad = []
    if len(exp) != len(got):
        bad.append(f"{len(exp)} records expected, {len(got)} found")
    for e, g in zip(exp, got):
        name = f"{'table' if e['tag'] == 1 else 'lookup'} kind {e['kind']} node {e['node']} #{e['which']}"
        for f in ("tag", "kind", "size", "aux", "aux2", "node", "which", "nl", "nc"):
            if e[f] != g[f]:
                bad.append(f"{name}: {f} {e[f]} != {g[f]}")
        if e["cols"].shape != g["cols"].shape:
            bad.append(f"{name}: shape {e['cols'].shape} != {g['cols'].shape}")
        elif not np.array_equal(e["cols"], g["cols"]):
            c, r = np.argwhere(e["cols"] != g["cols"])[0]
            bad.append(f"{name}: column {c} row {r}: {e['cols'][c, r]} != {g['cols'][c, r]}")
    return bad


MODELS = {
    "mlp": lambda: M.mlp(2, 64, config=7),                          # Requant (clamping + range), Relu
    "cnn": lambda: M.cnn_tiny(),                                    # + MaxPool difference columns, conv requants
    "gelu": lambda: M.gelu_mlp(16, config=31),                      # + the GELU table at the scaled input
    "layernorm": lambda: M.layernorm_mlp(4, 8, 16, config=33),      # + the inverse-square-root table, the scaled top chunk
    "softmax": lambda: M.softmax_only(2, 4, config=35, in_scale=8.0 / 127.0),             # + the exponential, error and zero tables, the late row-sum column
    "mha": lambda: M.mha_block(4, 8, 2, 4, config=37),              # the softmax inside an Mha node, QKV / ConcatMatMul around it
}


@pytest.mark.parametrize("name", sorted(MODELS))
def test_tables_and_witness_columns_of_both_cpp_copies_equal_the_numpy_writing(harness, tmp_path, name):
    mb = MODELS[name]()
    x = mb.input()
    recs = dump(harness, mb, x, tmp_path)
    exp = expected(mb, x)
    assert sum(1 for e in exp if e["tag"] == 1) >= 2 and sum(1 for e in exp if e["tag"] == 2) >= 2
    for src, who in ((0, "product csrc/zkml.h"), (1, "oracle oracle/zkml.hpp")):
        bad = differences(exp, recs[src])
        assert not bad, f"{who}: " + "; ".join(bad[:6])


def test_the_comparison_notices_one_changed_table_row_and_one_changed_witness_value(harness, tmp_path, monkeypatch):
    """the sensitivity the verdict asked for, shown from the numpy side: one row of the GELU look-up function off by one, one maxpool difference off by one —
    each makes BOTH C++ copies differ from the numpy writing in exactly the table / column concerned"""
    mb = MODELS["gelu"]()
    x = mb.input()
    recs = dump(harness, mb, x, tmp_path)
    real = M.gelu_table_output
    monkeypatch.setattr(M, "gelu_table_output", lambda s: real(s) + (1 if s == 17 else 0))
    bad = differences(expected(mb, x), recs[0])
    monkeypatch.setattr(M, "gelu_table_output", real)
    assert bad and all("table kind 1" in b or "lookup kind 1" in b for b in bad), bad
    mb = MODELS["cnn"]()
    x = mb.input()
    recs = dump(harness, mb, x, tmp_path)
    real_mp = LI.WITNESS[M.L_MAXPOOL]

    def skewed(col, node, l, v):
        c, h, w = l["pin"]
        t = np.asarray(v, dtype=np.int64).reshape(c, h // 2, 2, w // 2, 2)
        out = t.max(axis=(2, 4))
        diffs = [(out - t[:, :, dy, :, dx]).reshape(-1) for dy, dx in ((0, 0), (0, 1), (1, 0), (1, 1))]  # (dy, dx) of the middle two columns swapped
        col.add(node, 0, dict(kind=LI.K_RANGE), diffs, committed=diffs + [out.reshape(-1)])
        return out.reshape(-1)
    monkeypatch.setitem(LI.WITNESS, M.L_MAXPOOL, skewed)
    for src in (0, 1):
        bad = differences(expected(mb, x), recs[src])
        assert bad and all("lookup kind 2" in b for b in bad), bad
    monkeypatch.setitem(LI.WITNESS, M.L_MAXPOOL, real_mp)
