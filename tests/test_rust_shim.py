"""the Rust shim crates exist as SOURCE (no toolchain here) — what can be checked mechanically is checked: every
`extern "C"` declaration of rust/deep-prove-hip-sys against the prototype of the same name in include/deep_prove_hip.h (name, arity,
every argument type, the return type; both directions), every dp_* symbol of the built library against both, and every FFI call made
by rust/basefold-hip and the sumcheck patch against the declarations (function exists, number of arguments)."""
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "deep_prove_hip.h")
SYS = os.path.join(ROOT, "rust", "deep-prove-hip-sys", "src", "lib.rs")
CALLERS = [os.path.join(ROOT, "rust", "basefold-hip", "src", "lib.rs"), os.path.join(ROOT, "rust", "sumcheck-hip-patch", "prover_hip.rs"),
           os.path.join(ROOT, "rust", "zkml-hip-patch", "hip_blob.rs")]

SCALARS = {"int32_t": "i32", "uint32_t": "u32", "int64_t": "i64", "uint64_t": "u64", "size_t": "usize", "double": "f64", "uint8_t": "u8",
           "char": "c_char", "void": "c_void"}


def c_type_to_rust(t):
    """`const dp_buf* const*` -> `*const *const dp_buf`: read the C declarator right to left"""
    toks = re.findall(r"\*|\w+", t)
    base = [x for x in toks if x not in ("const", "*") and x != "struct"]
    assert len(base) == 1, t
    out = SCALARS.get(base[0], base[0])
    # walk the tokens: a `const` qualifies what is to its left (or, leading, the base type)
    levels = []  # constness of: base, then of each pointer
    i, cur_const = 0, False
    seen_base = False
    for tok in toks:
        if tok == "const":
            cur_const = True
        elif tok == "*":
            levels.append(cur_const)
            cur_const = False
        else:
            seen_base = True
    levels.append(cur_const)  # constness of the outermost object (irrelevant for a by-value parameter)
    for is_const in levels[:-1]:
        out = ("*const " if is_const else "*mut ") + out
    return out


def header_protos():
    src = re.sub(r"/\*.*?\*/", "", open(HEADER).read(), flags=re.S)
    src = re.sub(r"^\s*#.*$", "", src, flags=re.M)
    protos = {}
    for stmt in src.split(";"):
        stmt = " ".join(stmt.split())
        m = re.match(r"^(?:extern \"C\" \{ )?(.*?)\b(dp_\w+)\s*\((.*)\)$", stmt)
        if not m or "typedef" in stmt:
            continue
        ret, name, args = m.group(1).strip(), m.group(2), m.group(3).strip()
        params = []
        if args and args != "void":
            for a in args.split(","):
                a = a.strip()
                arr = re.search(r"\[\d*\]$", a)
                if arr:
                    a = a[:arr.start()].strip()
                ty = re.match(r"(.*?)(\w+)$", a).group(1).strip()
                params.append(c_type_to_rust(ty + ("*" if arr else "")))
        protos[name] = (c_type_to_rust(ret) if ret != "void" else None, params)
    return protos


def rust_externs():
    src = open(SYS).read()
    block = re.search(r'extern "C" \{(.*?)\n\}', src, flags=re.S).group(1)
    out = {}
    for m in re.finditer(r"pub fn (\w+)\((.*?)\)(?:\s*->\s*([^;]+))?;", block):
        name, args, ret = m.group(1), m.group(2).strip(), (m.group(3) or "").strip() or None
        params = [a.split(":", 1)[1].strip() for a in args.split(",")] if args else []
        out[name] = (ret, params)
    return out


def test_extern_block_matches_the_header_in_both_directions():
    h, r = header_protos(), rust_externs()
    assert len(h) >= 60, f"header parser found only {len(h)} prototypes"
    assert set(h) == set(r), f"only in the header: {sorted(set(h) - set(r))}; only in the extern block: {sorted(set(r) - set(h))}"
    for name in sorted(h):
        assert h[name][1] == r[name][1], f"{name}: header arguments {h[name][1]} != extern {r[name][1]}"
        assert h[name][0] == r[name][0], f"{name}: header returns {h[name][0]}, extern {r[name][0]}"


def test_status_codes_match_the_header():
    hdr, rs = open(HEADER).read(), open(SYS).read()
    for name, val in re.findall(r"#define (DP_\w+) \(?(-?\d+)\)?", hdr):
        m = re.search(rf"pub const {name}: i32 = (-?\d+);", rs)
        assert m and int(m.group(1)) == int(val), name


def test_library_exports_what_both_declare():
    import ctypes
    lib = os.path.join(ROOT, "deep-prove_amd", "libdeepprove_hip.so")
    if not os.path.exists(lib):
        pytest.skip("library not built")
    so = ctypes.CDLL(lib)
    for name in rust_externs():
        assert hasattr(so, name), f"{name} is declared but not exported by libdeepprove_hip.so"


def _calls(src):
    """(name, number of arguments) of every `sys::dp_*(...)` call, arguments counted at nesting depth 0"""
    for m in re.finditer(r"sys::(dp_\w+)\s*\(", src):
        i, depth, n, any_tok = m.end(), 0, 0, False
        while True:
            c = src[i]
            if c in "([{":
                depth += 1
            elif c in ")]}":
                if depth == 0:
                    break
                depth -= 1
            elif c == "," and depth == 0:
                n += 1
            elif not c.isspace():
                any_tok = True
            i += 1
        yield m.group(1), (n + 1 if any_tok else 0)


@pytest.mark.parametrize("path", CALLERS)
def test_ffi_calls_of_the_shim_crates_fit_the_declarations(path):
    decl = rust_externs()
    src = open(path).read()
    seen = list(_calls(src))
    assert seen, path
    for name, nargs in seen:
        assert name in decl, f"{os.path.basename(path)} calls {name}, which the extern block does not declare"
        assert nargs == len(decl[name][1]), f"{os.path.basename(path)}: {name} called with {nargs} arguments, declared with {len(decl[name][1])}"


def test_trait_surface_is_complete():
    """every required fn of mpcs::PolynomialCommitmentScheme (mpcs/src/lib.rs:111-226) has an implementation in BasefoldHip"""
    src = open(CALLERS[0]).read()
    for fn in ("setup", "trim", "commit", "write_commitment", "get_pure_commitment", "trivial_num_vars", "batch_commit", "open", "batch_open",
               "simple_batch_open", "verify", "batch_verify", "simple_batch_verify"):
        assert re.search(rf"\bfn {fn}\(", src), fn
    for ty in ("Param", "ProverParam", "VerifierParam", "CommitmentWithWitness", "Commitment", "CommitmentChunk", "Proof"):
        assert re.search(rf"type {ty} =", src), ty


def test_model_blob_writer_uses_the_layer_kinds_of_the_library_and_covers_the_layer_enum():
    """rust/zkml-hip-patch/hip_blob.rs (seam 3: Model<Element> -> the blob of dp_model_setup): its KIND_* constants equal enum LayerKind of
    csrc/proof.h (names and numbers, both directions), and it has a match arm for every variant of the reference's Layer enum
    (zkml/src/layers/mod.rs:66-93; the variant list is restated here: /root/reference is not on the GPU box)"""
    src = open(os.path.join(ROOT, "rust", "zkml-hip-patch", "hip_blob.rs")).read()
    rust_kinds = {m.group(1): int(m.group(2)) for m in re.finditer(r"pub const KIND_(\w+): i64 = (\d+);", src)}
    hdr = open(os.path.join(ROOT, "deep-prove_amd", "csrc", "proof.h")).read()
    enum = re.search(r"enum LayerKind \{(.*?)\};", hdr, re.S).group(1)
    c_kinds = {m.group(1): int(m.group(2)) for m in re.finditer(r"L_(\w+) = (\d+)", enum)}
    assert rust_kinds == c_kinds and len(c_kinds) == 18
    from deep_prove_amd import models as M
    for name, val in c_kinds.items():  # the Python writer of the golden fixtures uses the same numbers
        assert getattr(M, "L_" + name) == val
    for variant in ("Dense", "MatMul", "Convolution", "SchoolBookConvolution", "Activation", "Requant", "Pooling", "Flatten", "QKV", "Mha", "ConcatMatMul",
                    "LayerNorm", "Softmax", "Add", "Reshape", "Embeddings", "Positional", "Logits"):
        assert re.search(rf"Layer::{variant}\b", src), variant
    for kind in rust_kinds:  # every kind is written by some arm
        assert len(re.findall(rf"\bKIND_{kind}\b", src)) >= 2, kind


def test_model_blob_writer_compiles_against_its_own_error_type_and_lists_its_accessors():
    """`model_to_blob` returns Result<_, BlobError>, so no arm may build an anyhow error with `?` (there is no From<anyhow::Error>); and
    every getter hip_blob.rs calls that the reference's structs do not have is an explicit `+` hunk of rust/zkml-hip-patch/accessors.patch"""
    src = open(os.path.join(ROOT, "rust", "zkml-hip-patch", "hip_blob.rs")).read()
    assert "anyhow!" not in src and "use anyhow" not in src
    patch = open(os.path.join(ROOT, "rust", "zkml-hip-patch", "accessors.patch")).read()
    added = set(re.findall(r"^\+\s+pub\(crate\) fn (\w+)\(", patch, re.M))
    for fn in ("quant_data", "multiplier", "operand", "multipliers", "is_transposed_b", "inner_and_output_dims", "padded_filter", "padded_input_side",
               "unpadded_output_shape", "table", "padded_input_shapes_of", "padded_input_shape_of"):
        assert re.search(rf"\.{fn}\(", src), fn
        assert fn in added, fn
    for line in patch.splitlines():  # hunks only add
        assert not (line.startswith("-") and not line.startswith("---")), line
