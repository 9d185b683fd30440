#!/bin/bash
# The pin kit: ONE command for a machine that has what this repository's container lacks — a Rust toolchain (nightly-2025-05-22, the reference's
# rust-toolchain.toml), network or a vendored registry for the Plonky3 git dependencies (rev f37dc2a5), and the reference checked out.
#   usage: tools/pin_with_cargo.sh /path/to/deep-prove [--with-shims] [--vendor DIR | --make-vendor DIR]
# 1. builds tools/pin/reference_pin (a standalone package that reaches the reference's crates by path) and runs it: the REFERENCE's Poseidon2 permutation,
#    compress, BasicTranscript challenges, a prove_parallel proof, a Basefold commitment root (L0-L2) and — round 6 — logup-GKR batch_prove of a lookup and of
#    its table, Basefold::batch_open of three polynomials of mixed size and field, and two whole Prover::prove runs (Dense 128 x 128 = BASELINE config 1;
#    one Dense + Requant + ReLU block) built with the reference's own Model API from the SplitMix64 tensors of deep-prove_amd/models.py (L3-L4), every object
#    as the hex of rmp_serde::to_vec_named -> tests/golden/reference_pin.json. The header of src/main.rs says which rows of SURVEY §8(a) each key pins;
# 2. runs tests/test_reference_pin.py, which replays the same inputs through the oracle and compares value by value and object by object (field names, order and
#    values of the decoded msgpack, then the bytes) — the day this passes, SURVEY §8c's "parity unpinned" is closed for every row of §8(a) except a21, and the
#    msgpack conventions of deep-prove_amd/wire.py are checked against real reference bytes (§8 f2);
# Offline machines: `--make-vendor DIR` (on a machine WITH network) runs `cargo vendor` for the pin package — the Plonky3 crates at rev f37dc2a5 (p3-field,
#    p3-goldilocks, p3-poseidon2, p3-symmetric, p3-challenger, p3-mds, p3-dft, p3-matrix, p3-util, p3-maybe-rayon: Cargo.lock:6742-6858 of the reference), ceno's
#    `goldilocks` @29a15d1 and the crates.io closure of the reference's workspace — into DIR together with the `.cargo/config.toml` that points cargo at it;
#    `--vendor DIR` on the offline machine builds against that directory with `--offline`.
# 3. --with-shims: cargo check of rust/{deep-prove-hip-sys,basefold-hip} against the reference's workspace (seams 1 + 2 compiled for the first time).
set -euo pipefail
REF=$(realpath "${1:?usage: tools/pin_with_cargo.sh /path/to/deep-prove [--with-shims]}")
ROOT=$(cd "$(dirname "$0")/.." && pwd)
command -v cargo > /dev/null || { echo "pin_with_cargo: no cargo on PATH (the reference pins nightly-2025-05-22)"; exit 2; }
[ -f "$REF/ff_ext/src/lib.rs" ] || { echo "pin_with_cargo: $REF is not a deep-prove checkout"; exit 2; }
W=$(mktemp -d)
cp -r "$ROOT/tools/pin/reference_pin" "$W/reference_pin"
sed -i "s#REFERENCE_DIR#$REF#g" "$W/reference_pin/Cargo.toml"
cp "$REF/rust-toolchain.toml" "$W/reference_pin/" 2> /dev/null || true
OFFLINE=""
for ((i = 2; i <= $#; i++)); do
  if [ "${!i}" = "--make-vendor" ]; then j=$((i + 1)); V=$(realpath -m "${!j}"); mkdir -p "$V"
    (cd "$W/reference_pin" && cargo vendor "$V/vendor" > "$V/cargo-config.toml"); echo "vendored into $V (copy it to the offline machine and pass --vendor $V there)"; exit 0; fi
  if [ "${!i}" = "--vendor" ]; then j=$((i + 1)); V=$(realpath "${!j}"); mkdir -p "$W/reference_pin/.cargo"
    sed "s#directory = .*#directory = \"$V/vendor\"#" "$V/cargo-config.toml" > "$W/reference_pin/.cargo/config.toml"; OFFLINE="--offline"; fi
done
(cd "$W/reference_pin" && cargo run --release $OFFLINE) | tail -1 > "$ROOT/tests/golden/reference_pin.json"
echo "wrote tests/golden/reference_pin.json:"; head -c 400 "$ROOT/tests/golden/reference_pin.json"; echo
(cd "$ROOT" && python -m pytest tests/test_reference_pin.py -q)
if [[ " $* " == *" --with-shims "* ]]; then
  export DEEP_PROVE_HIP_LIB_DIR="$ROOT/deep-prove_amd"
  for c in deep-prove-hip-sys basefold-hip; do
    cp -r "$ROOT/rust/$c" "$W/$c"; sed -i "s#\.\./\.\./reference#$REF#g; s#/root/reference#$REF#g" "$W/$c/Cargo.toml"
    (cd "$W/$c" && cargo check) || { echo "pin_with_cargo: $c does not compile yet (never built before: see rust/README.md)"; exit 3; }
  done
fi
