#!/bin/bash
# The pin kit: ONE command for a machine that has what this repository's container lacks — a Rust toolchain (nightly-2025-05-22, the reference's
# rust-toolchain.toml), network or a vendored registry for the Plonky3 git dependencies (rev f37dc2a5), and the reference checked out.
#   usage: tools/pin_with_cargo.sh /path/to/deep-prove [--with-shims]
# 1. builds tools/pin/reference_pin (a standalone package that reaches the reference's crates by path) and runs it: the REFERENCE's Poseidon2 permutation,
#    compress, BasicTranscript challenges, a prove_parallel proof, a Basefold commitment root and their rmp_serde bytes on fixed inputs
#    -> tests/golden/reference_pin.json;
# 2. runs tests/test_reference_pin.py, which replays the same inputs through the oracle (and through libdeepprove_hip.so when a GPU is present) and compares —
#    the day this passes, SURVEY §8c's "parity unpinned" is closed for layers L0-L2, and the msgpack conventions of deep-prove_amd/wire.py are checked against
#    real reference bytes;
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
(cd "$W/reference_pin" && cargo run --release) | tail -1 > "$ROOT/tests/golden/reference_pin.json"
echo "wrote tests/golden/reference_pin.json:"; head -c 400 "$ROOT/tests/golden/reference_pin.json"; echo
(cd "$ROOT" && python -m pytest tests/test_reference_pin.py -q)
if [ "${2:-}" = "--with-shims" ]; then
  export DEEP_PROVE_HIP_LIB_DIR="$ROOT/deep-prove_amd"
  for c in deep-prove-hip-sys basefold-hip; do
    cp -r "$ROOT/rust/$c" "$W/$c"; sed -i "s#\.\./\.\./reference#$REF#g; s#/root/reference#$REF#g" "$W/$c/Cargo.toml"
    (cd "$W/$c" && cargo check) || { echo "pin_with_cargo: $c does not compile yet (never built before: see rust/README.md)"; exit 3; }
  done
fi
