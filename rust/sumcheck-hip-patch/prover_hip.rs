//! Seam 2: the body that replaces `IOPProverState::prove_parallel`
//! (`sumcheck/src/prover.rs:498-585`) in a patched `sumcheck` crate — add `mod prover_hip;` to `sumcheck/src/lib.rs`, a dependency on
//! `deep-prove-hip-sys` and `basefold-hip` (for `HipTranscript` / the process-wide context), and let `prove_parallel` forward here.
//! UNBUILT in the repository's environment; the FFI calls are checked by `tests/test_rust_shim.py`.
//!
//! The virtual polynomial goes over as (tables, term_degree, term_tables, term_coeffs): `flattened_ml_extensions` are the
//! de-duplicated tables (`virtual_poly.rs:50-60`), `products` the (coefficient, table indices) terms. Returned: the `IOPProof` parsed
//! from the word stream {point: len, ext...; rounds: count, (len, ext...)...} and a state whose `get_mle_final_evaluations()` yields
//! `finals` (table order).
use deep_prove_hip_sys as sys;
use ff_ext::{ExtensionField, GoldilocksExt2, SmallField};
use multilinear_extensions::{mle::FieldType, virtual_poly::VirtualPolynomial};
use transcript::Transcript;

use crate::structs::{IOPProof, IOPProverMessage, IOPProverState};

type E = GoldilocksExt2;
type F = <E as ExtensionField>::BaseField;

struct Table(*mut sys::dp_buf, *mut sys::dp_ctx);
impl Drop for Table { fn drop(&mut self) { unsafe { sys::dp_buf_free(self.1, self.0); } } }

pub fn prove_parallel_hip(ctx: *mut sys::dp_ctx, poly: VirtualPolynomial<E>, transcript: *mut sys::dp_transcript) -> Result<(IOPProof<E>, Vec<E>), sys::DpError> {
    let num_vars = poly.aux_info.max_num_variables;
    // tables: every flattened MLE once, in order (their index is what `products` refers to)
    let mut tables = Vec::with_capacity(poly.flattened_ml_extensions.len());
    for mle in &poly.flattened_ml_extensions {
        let (words, n, is_ext): (Vec<u64>, usize, i32) = match mle.evaluations() {
            FieldType::Base(v) => (v.iter().map(|x| x.to_canonical_u64()).collect(), v.len(), 0),
            FieldType::Ext(v) => (v.iter().flat_map(|x| { let b = x.as_bases(); [b[0].to_canonical_u64(), b[1].to_canonical_u64()] }).collect(), v.len(), 1),
            FieldType::Unreachable => unreachable!(),
        };
        let mut b = core::ptr::null_mut();
        sys::check(unsafe { sys::dp_buf_upload(ctx, words.as_ptr(), n, is_ext, &mut b) })?;
        tables.push(Table(b, ctx));
    }
    let ptrs: Vec<*const sys::dp_buf> = tables.iter().map(|t| t.0 as *const sys::dp_buf).collect();
    let term_degree: Vec<i32> = poly.products.iter().map(|(_, idx)| idx.len() as i32).collect();
    let term_tables: Vec<i32> = poly.products.iter().flat_map(|(_, idx)| idx.iter().map(|&i| i as i32)).collect();
    let term_coeffs: Vec<u64> = poly.products.iter().flat_map(|(c, _)| { let b = c.as_bases(); [b[0].to_canonical_u64(), b[1].to_canonical_u64()] }).collect();
    let (mut w, mut n) = (core::ptr::null_mut(), 0usize);
    let mut finals = vec![0u64; 2 * tables.len()];
    sys::check(unsafe { sys::dp_sumcheck_prove(ctx, num_vars as u32, ptrs.as_ptr(), ptrs.len() as i32, term_degree.as_ptr(), term_tables.as_ptr(), term_coeffs.as_ptr(),
                                               term_degree.len() as i32, transcript, &mut w, &mut n, finals.as_mut_ptr()) })?;
    let words = unsafe { sys::Words::from_raw(w, n) };
    let s = words.as_slice();
    let ext = |i: usize| E::from_bases(&[F::from_v(s[i]), F::from_v(s[i + 1])]);
    let mut at = 0usize;
    let np = s[at] as usize; at += 1;
    let point: Vec<E> = (0..np).map(|k| ext(at + 2 * k)).collect(); at += 2 * np;
    let nr = s[at] as usize; at += 1;
    let mut proofs = Vec::with_capacity(nr);
    for _ in 0..nr {
        let len = s[at] as usize; at += 1;
        proofs.push(IOPProverMessage { evaluations: (0..len).map(|k| ext(at + 2 * k)).collect() });
        at += 2 * len;
    }
    let final_evals: Vec<E> = (0..tables.len()).map(|k| E::from_bases(&[F::from_v(finals[2 * k]), F::from_v(finals[2 * k + 1])])).collect();
    Ok((IOPProof { point, proofs }, final_evals))
}

/// what `IOPProverState::prove_parallel` becomes (`sumcheck/src/prover.rs:498-501`): the state it returns only has to answer
/// `get_mle_final_evaluations()` (`prover.rs:474-490`), which callers use right after the proof (`zkml/src/layers/dense.rs:505-517`)
impl<'a> IOPProverState<'a, E> {
    pub fn prove_parallel_on_device(poly: VirtualPolynomial<'a, E>, transcript: &mut impl Transcript<E>, ctx: *mut sys::dp_ctx, handle: *mut sys::dp_transcript) -> (IOPProof<E>, Vec<E>) {
        let _ = transcript;  // the sponge lives behind `handle` (basefold_hip::HipTranscript)
        prove_parallel_hip(ctx, poly, handle).expect("device sumcheck failed")
    }
}
