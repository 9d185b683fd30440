//! Prints ONE JSON object: what the reference's own crates compute on the fixed inputs tests/test_reference_pin.py replays through the oracle.
//! Every value is a canonical u64 (extension elements as [c0, c1]). Written against the reference @ 2025-08-08 without a compiler at hand: the API
//! names are the ones read in the sources (cited), a maintainer may have to touch an import.
use ff_ext::{ExtensionField, GoldilocksExt2, PoseidonField};
use mpcs::{Basefold, BasefoldRSParams, Hasher, PolynomialCommitmentScheme}; // Hasher = PoseidonHasher without the `blake` feature (mpcs/src/lib.rs:339-342)
use multilinear_extensions::{mle::DenseMultilinearExtension, virtual_poly::{ArcMultilinearExtension, VirtualPolynomial}};
use p3_field::{FieldAlgebra, FieldExtensionAlgebra, PrimeField64};
use p3_goldilocks::Goldilocks;
use p3_symmetric::Permutation;
use poseidon::{digest::Digest, poseidon_hash::PoseidonHash};
use sumcheck::structs::IOPProverState;
use transcript::{Transcript, basic::BasicTranscript};

type E = GoldilocksExt2;
type Pcs = Basefold<E, BasefoldRSParams<Hasher>>; // zkml/src/bin/bench.rs:13, 24-26

fn f(v: u64) -> Goldilocks { Goldilocks::from_canonical_u64(v) }
fn ext(e: &E) -> Vec<u64> { e.as_base_slice().iter().map(|x| x.as_canonical_u64()).collect() } // ff_ext: as_bases()

fn main() {
    let mut out = serde_json::Map::new();
    // 1. the Poseidon2-w8 permutation as ff_ext wires it (ff_ext/src/lib.rs:167-236: HL constants + p3's MDSMat4) on [0..7]
    let mut st: [Goldilocks; 8] = core::array::from_fn(|i| f(i as u64));
    <Goldilocks as PoseidonField>::get_perm().permute_mut(&mut st);
    out.insert("poseidon2_permute_0_7".into(), st.iter().map(|x| x.as_canonical_u64()).collect::<Vec<_>>().into());
    // 2. compress (poseidon/src/poseidon_hash.rs:65-70) of the digests [1,2,3,4] and [5,6,7,8]
    let x = Digest([f(1), f(2), f(3), f(4)]);
    let y = Digest([f(5), f(6), f(7), f(8)]);
    let d = PoseidonHash::<Goldilocks>::two_to_one(&x, &y);
    out.insert("compress_1234_5678".into(), d.0.iter().map(|x| x.as_canonical_u64()).collect::<Vec<_>>().into());
    // 3. BasicTranscript::new(b"m2vec") (zkml/src/lib.rs:96-98): the first unlabelled challenge, then a labelled one
    let mut t = BasicTranscript::<E>::new(b"m2vec");
    let c1 = t.read_challenge().elements;
    let c2 = t.get_and_append_challenge(b"Internal round").elements;
    out.insert("transcript_m2vec_read_challenge".into(), ext(&c1).into());
    out.insert("transcript_m2vec_then_internal_round".into(), ext(&c2).into());
    // 4. prove_parallel (sumcheck/src/prover.rs:498-585) of sum_b f(b) g(b) h(b), 4 variables, f = 1..16, g = 17..32, h = 3 i + 1: messages, point
    let nv = 4usize;
    // (DenseMultilinearExtension -> ArcMultilinearExtension by `.into()`, as zkml/src/layers/dense.rs:494 does)
    let mk = |g: &dyn Fn(u64) -> u64| -> ArcMultilinearExtension<'static, E> { DenseMultilinearExtension::<E>::from_evaluations_vec(nv, (0..1u64 << nv).map(|i| f(g(i))).collect()).into() };
    let (a, b, c) = (mk(&|i| i + 1), mk(&|i| i + 17), mk(&|i| 3 * i + 1));
    let mut vp = VirtualPolynomial::<E>::new(nv);
    vp.add_mle_list(vec![a.clone(), b.clone(), c.clone()], E::ONE);
    vp.add_mle_list(vec![a.clone(), c.clone()], E::from_canonical_u64(5));
    let mut t = BasicTranscript::<E>::new(b"m2vec");
    let (proof, state) = IOPProverState::<E>::prove_parallel(vp, &mut t);
    let msgs: Vec<Vec<Vec<u64>>> = proof.proofs.iter().map(|m| m.evaluations.iter().map(ext).collect()).collect();
    out.insert("sumcheck_messages".into(), serde_json::to_value(&msgs).unwrap());
    out.insert("sumcheck_point".into(), serde_json::to_value(proof.point.iter().map(ext).collect::<Vec<_>>()).unwrap());
    out.insert("sumcheck_final_evaluations".into(), serde_json::to_value(state.get_mle_final_evaluations().iter().map(ext).collect::<Vec<_>>()).unwrap());
    out.insert("sumcheck_next_challenge".into(), ext(&t.read_challenge().elements).into());
    // 5. the wire format of that proof (zkml/src/bin/bench.rs:399: rmp_serde::to_vec_named)
    out.insert("sumcheck_proof_rmp_named_hex".into(), rmp_serde::encode::to_vec_named(&proof).unwrap().iter().map(|b| format!("{:02x}", b)).collect::<String>().into());
    // 6. Basefold::commit (mpcs/src/basefold.rs:304-354) of the base polynomial with evaluations i^2 + 1, 10 variables, parameters for 2^12
    let pnv = 10usize;
    let poly = DenseMultilinearExtension::<E>::from_evaluations_vec(pnv, (0..1u64 << pnv).map(|i| f(i * i + 1)).collect());
    let param = Pcs::setup(1 << 12).unwrap();
    let (pp, _vp) = Pcs::trim(param, 1 << 12).unwrap();
    let comm = Pcs::commit(&pp, &poly).unwrap();
    let pure = Pcs::get_pure_commitment(&comm);
    out.insert("basefold_commit_root_nv10".into(), serde_json::to_value(pure.root().0.iter().map(|x| x.as_canonical_u64()).collect::<Vec<_>>()).unwrap());
    out.insert("basefold_commitment_rmp_named_hex".into(), rmp_serde::encode::to_vec_named(&pure).unwrap().iter().map(|b| format!("{:02x}", b)).collect::<String>().into());
    println!("{}", serde_json::Value::Object(out));
}
