//! Raw FFI to `libdeepprove_hip.so` — the C ABI of `include/deep_prove_hip.h`, one declaration per prototype.
//!
//! UNBUILT in this repository's environment (no Rust toolchain). `tests/test_rust_shim.py` parses this file and the header
//! independently and fails on any name, arity or type mismatch, in either direction.
//!
//! Conventions of the ABI (header, top comment): every function returns an `i32` status (0 = ok, negative = `DP_ERR_*`) and
//! never unwinds; `dp_last_error()` is the message of the calling thread's last failure; field elements are canonical
//! little-endian `u64` (< p = 2^64 - 2^32 + 1), an extension element is `[c0, c1]` of `c0 + c1 X`, `X^2 = 7`; buffers returned
//! through `*mut *mut u64` belong to the library and go back with `dp_free`. There is NO CPU fallback: `dp_ctx_create` fails
//! with `DP_ERR_NODEVICE` without an MI355X.
#![allow(non_camel_case_types)]
use core::ffi::{c_char, c_void};

pub const DP_OK: i32 = 0;
pub const DP_ERR_ARG: i32 = -1;
pub const DP_ERR_OOM: i32 = -2;
pub const DP_ERR_HIP: i32 = -3;
pub const DP_ERR_SHAPE: i32 = -4;
pub const DP_ERR_VERIFY: i32 = -5;
pub const DP_ERR_NODEVICE: i32 = -6;

macro_rules! opaque { ($($n:ident),*) => { $( #[repr(C)] pub struct $n { _p: [u8; 0], _m: core::marker::PhantomData<(*mut u8, core::marker::PhantomPinned)> } )* } }
opaque!(dp_ctx, dp_buf, dp_transcript, dp_commit, dp_model, dp_sc_session, dp_dist, dp_batch_commit, dp_async, dp_ticket);

extern "C" {
    pub fn dp_last_error() -> *const c_char;
    pub fn dp_free(p: *mut c_void);
    pub fn dp_ctx_create(device_id: i32, out_: *mut *mut dp_ctx) -> i32;
    pub fn dp_ctx_destroy(ctx: *mut dp_ctx) -> i32;
    pub fn dp_ctx_name(ctx: *const dp_ctx) -> *const c_char;
    pub fn dp_ctx_set_throughput_mode(ctx: *mut dp_ctx, on: i32) -> i32;
    pub fn dp_async_create(ctx: *mut dp_ctx, max_in_flight: i32, worker_arena_bytes: usize, out_: *mut *mut dp_async) -> i32;
    pub fn dp_async_destroy(a: *mut dp_async) -> i32;
    pub fn dp_ctx_route_to_engine(ctx: *mut dp_ctx, engine: *mut dp_async) -> i32;
    pub fn dp_async_stats(a: *mut dp_async, calls: *mut usize, groups: *mut usize, merged_calls: *mut usize, workers: *mut usize) -> i32;
    pub fn dp_pcs_commit_submit(a: *mut dp_async, poly: *const dp_buf, ticket: *mut *mut dp_ticket) -> i32;
    pub fn dp_pcs_commit_host_submit(a: *mut dp_async, words: *const u64, n: usize, is_ext: i32, ticket: *mut *mut dp_ticket) -> i32;
    pub fn dp_mle_fix_high_submit(a: *mut dp_async, matrix: *const dp_buf, rows: usize, cols: usize, point: *const u64, ticket: *mut *mut dp_ticket) -> i32;
    pub fn dp_mle_eval_submit(a: *mut dp_async, f: *const dp_buf, point: *const u64, k: u32, ticket: *mut *mut dp_ticket) -> i32;
    pub fn dp_ticket_buf(t: *mut dp_ticket, out_: *mut *mut dp_buf) -> i32;
    pub fn dp_sumcheck_prove_submit(a: *mut dp_async, num_vars: u32, tables: *const *const dp_buf, ntables: i32, term_degree: *const i32, term_tables: *const i32, term_coeffs: *const u64, nterms: i32, t: *mut dp_transcript, ticket: *mut *mut dp_ticket) -> i32;
    pub fn dp_logup_prove_submit(a: *mut dp_async, columns: *const *const dp_buf, ncols: i32, cols_per_instance: i32, multiplicities: *const dp_buf, constant_challenge: *const u64, column_separation_challenge: *const u64, t: *mut dp_transcript, ticket: *mut *mut dp_ticket) -> i32;
    pub fn dp_pcs_batch_open_submit(a: *mut dp_async, comms: *const *const dp_commit, n: i32, points_flat: *const u64, evals: *const u64, t: *mut dp_transcript, ticket: *mut *mut dp_ticket) -> i32;
    pub fn dp_poll(t: *mut dp_ticket) -> i32;
    pub fn dp_wait(t: *mut dp_ticket) -> i32;
    pub fn dp_ticket_words(t: *mut dp_ticket, which: i32, words: *mut *mut u64, nwords: *mut usize) -> i32;
    pub fn dp_ticket_values(t: *mut dp_ticket, values: *mut u64, nvalues: usize) -> i32;
    pub fn dp_ticket_commit(t: *mut dp_ticket, out_: *mut *mut dp_commit, root: *mut u64) -> i32;
    pub fn dp_ticket_free(t: *mut dp_ticket) -> i32;
    pub fn dp_profile_enable(ctx: *mut dp_ctx, on: i32) -> i32;
    pub fn dp_profile_report(ctx: *mut dp_ctx, json: *mut *mut c_char) -> i32;
    pub fn dp_probe_compress_rate(ctx: *mut dp_ctx, nodes: usize, reps: i32, per_second: *mut f64) -> i32;
    pub fn dp_buf_from_i64(ctx: *mut dp_ctx, v: *const i64, n: usize, out_: *mut *mut dp_buf) -> i32;
    pub fn dp_buf_upload(ctx: *mut dp_ctx, words: *const u64, n: usize, is_ext: i32, out_: *mut *mut dp_buf) -> i32;
    pub fn dp_buf_download(ctx: *mut dp_ctx, buf: *const dp_buf, out_words: *mut u64) -> i32;
    pub fn dp_buf_len(buf: *const dp_buf) -> usize;
    pub fn dp_buf_is_ext(buf: *const dp_buf) -> i32;
    pub fn dp_buf_free(ctx: *mut dp_ctx, buf: *mut dp_buf) -> i32;
    pub fn dp_transcript_new(label: *const c_char) -> *mut dp_transcript;
    pub fn dp_transcript_free(t: *mut dp_transcript);
    pub fn dp_transcript_append_elements(t: *mut dp_transcript, base_elems: *const u64, n: usize) -> i32;
    pub fn dp_transcript_append_message(t: *mut dp_transcript, bytes: *const u8, n: usize) -> i32;
    pub fn dp_transcript_challenge(t: *mut dp_transcript, label: *const c_char, out_: *mut u64) -> i32;
    pub fn dp_eq_table(ctx: *mut dp_ctx, point: *const u64, k: u32, out_: *mut *mut dp_buf) -> i32;
    pub fn dp_mle_eval(ctx: *mut dp_ctx, f: *const dp_buf, point: *const u64, k: u32, out_: *mut u64) -> i32;
    pub fn dp_mle_fix_high(ctx: *mut dp_ctx, matrix: *const dp_buf, rows: usize, cols: usize, point: *const u64, out_: *mut *mut dp_buf) -> i32;
    pub fn dp_sumcheck_prove(ctx: *mut dp_ctx, num_vars: u32, tables: *const *const dp_buf, ntables: i32, term_degree: *const i32, term_tables: *const i32, term_coeffs: *const u64, nterms: i32, t: *mut dp_transcript, proof_words: *mut *mut u64, proof_nwords: *mut usize, finals: *mut u64) -> i32;
    pub fn dp_sumcheck_verify(num_vars: u32, max_degree: u32, claimed_sum: *const u64, proof_words: *const u64, proof_nwords: usize, t: *mut dp_transcript, point: *mut u64, expected_evaluation: *mut u64) -> i32;
    pub fn dp_sc_session_new(ctx: *mut dp_ctx, num_vars: u32, tables: *const *const dp_buf, ntables: i32, term_degree: *const i32, term_tables: *const i32, nterms: i32, out_: *mut *mut dp_sc_session) -> i32;
    pub fn dp_sc_session_round(s: *mut dp_sc_session, r_prev: *const u64, raw_out: *mut u64, nraw_ext: *mut usize) -> i32;
    pub fn dp_sc_session_finish(s: *mut dp_sc_session, r_last: *const u64, finals: *mut u64) -> i32;
    pub fn dp_sc_session_free(s: *mut dp_sc_session) -> i32;
    pub fn dp_dist_unique_id(id: *mut u8) -> i32;
    pub fn dp_dist_init(ctx: *mut dp_ctx, id: *const u8, rank: i32, world: i32, out_: *mut *mut dp_dist) -> i32;
    pub fn dp_dist_free(d: *mut dp_dist) -> i32;
    pub fn dp_sumcheck_prove_sharded(ctx: *mut dp_ctx, dist: *mut dp_dist, num_vars: u32, tables: *const *const dp_buf, ntables: i32, term_degree: *const i32, term_tables: *const i32, term_coeffs: *const u64, nterms: i32, t: *mut dp_transcript, proof_words: *mut *mut u64, proof_nwords: *mut usize, finals: *mut u64) -> i32;
    pub fn dp_sumcheck_prove_sharded_local(ctxs: *const *mut dp_ctx, world: i32, num_vars: u32, tables: *const *const dp_buf, ntables: i32, term_degree: *const i32, term_tables: *const i32, term_coeffs: *const u64, nterms: i32, transcripts: *const *mut dp_transcript, proof_words: *mut *mut u64, proof_nwords: *mut usize, finals: *mut u64) -> i32;
    pub fn dp_logup_prove(ctx: *mut dp_ctx, columns: *const *const dp_buf, ncols: i32, cols_per_instance: i32, multiplicities: *const dp_buf, constant_challenge: *const u64, column_separation_challenge: *const u64, t: *mut dp_transcript, proof_words: *mut *mut u64, proof_nwords: *mut usize) -> i32;
    pub fn dp_logup_verify(proof_words: *const u64, proof_nwords: usize, num_instances: i32, constant_challenge: *const u64, column_separation_challenge: *const u64, t: *mut dp_transcript, numerators: *mut u64, denominators: *mut u64, claims_words: *mut *mut u64, claims_nwords: *mut usize) -> i32;
    pub fn dp_pcs_setup(ctx: *mut dp_ctx, max_poly_size: usize) -> i32;
    pub fn dp_pcs_commit(ctx: *mut dp_ctx, poly: *const dp_buf, out_: *mut *mut dp_commit, root: *mut u64) -> i32;
    pub fn dp_pcs_commit_free(ctx: *mut dp_ctx, c: *mut dp_commit) -> i32;
    pub fn dp_pcs_commitment(c: *const dp_commit, root: *mut u64, num_vars: *mut u32, is_base: *mut i32) -> i32;
    pub fn dp_pcs_open(ctx: *mut dp_ctx, comm: *const dp_commit, point: *const u64, num_vars: u32, eval: *const u64, t: *mut dp_transcript, proof_words: *mut *mut u64, proof_nwords: *mut usize) -> i32;
    pub fn dp_pcs_verify(max_poly_size: usize, root: *const u64, num_vars: u32, is_base: i32, point: *const u64, eval: *const u64, proof_words: *const u64, proof_nwords: usize, t: *mut dp_transcript) -> i32;
    pub fn dp_pcs_batch_open(ctx: *mut dp_ctx, comms: *const *const dp_commit, n: i32, points_flat: *const u64, evals: *const u64, t: *mut dp_transcript, proof_words: *mut *mut u64, proof_nwords: *mut usize) -> i32;
    pub fn dp_pcs_batch_verify(max_poly_size: usize, roots: *const u64, num_vars: *const u32, is_base: *const i32, n: i32, points_flat: *const u64, evals: *const u64, proof_words: *const u64, proof_nwords: usize, t: *mut dp_transcript) -> i32;
    pub fn dp_pcs_batch_open_evals(ctx: *mut dp_ctx, comms: *const *const dp_commit, n_polys: i32, points_flat: *const u64, point_num_vars: *const u32, n_points: i32, eval_poly: *const u32, eval_point: *const u32, eval_values: *const u64, n_evals: i32, t: *mut dp_transcript, proof_words: *mut *mut u64, proof_nwords: *mut usize) -> i32;
    pub fn dp_pcs_batch_verify_evals(max_poly_size: usize, roots: *const u64, num_vars: *const u32, is_base: *const i32, n_polys: i32, points_flat: *const u64, point_num_vars: *const u32, n_points: i32, eval_poly: *const u32, eval_point: *const u32, eval_values: *const u64, n_evals: i32, proof_words: *const u64, proof_nwords: usize, t: *mut dp_transcript) -> i32;
    pub fn dp_pcs_batch_commit(ctx: *mut dp_ctx, polys: *const *const dp_buf, n: i32, out_: *mut *mut dp_batch_commit, root: *mut u64) -> i32;
    pub fn dp_pcs_batch_commit_free(ctx: *mut dp_ctx, c: *mut dp_batch_commit) -> i32;
    pub fn dp_pcs_simple_batch_open(ctx: *mut dp_ctx, comm: *const dp_batch_commit, point: *const u64, num_vars: u32, t: *mut dp_transcript, proof_words: *mut *mut u64, proof_nwords: *mut usize) -> i32;
    pub fn dp_pcs_simple_batch_verify(max_poly_size: usize, root: *const u64, num_vars: u32, is_base: i32, point: *const u64, evals: *const u64, n: i32, proof_words: *const u64, proof_nwords: usize, t: *mut dp_transcript) -> i32;
    pub fn dp_model_setup(ctx: *mut dp_ctx, model_blob: *const i64, nwords: usize, out_: *mut *mut dp_model) -> i32;
    pub fn dp_model_free(m: *mut dp_model) -> i32;
    pub fn dp_model_prove(m: *mut dp_model, input: *const i64, ninput: usize, proof_words: *mut *mut u64, proof_nwords: *mut usize, output: *mut i64, noutput: *mut usize, prove_ms: *mut f64) -> i32;
    pub fn dp_model_prove_batch(m: *mut dp_model, inputs: *const i64, nproofs: usize, ninput: usize, concurrency: i32, proof_words: *mut *mut u64, proof_nwords: *mut usize, outputs: *mut i64, noutput_cap: usize, noutput: *mut usize, wall_ms: *mut f64) -> i32;
    pub fn dp_model_infer_host(model_blob: *const i64, nwords: usize, input: *const i64, ninput: usize, output: *mut i64, noutput: *mut usize) -> i32;
    pub fn dp_host_poseidon2(state: *mut u64, force_scalar: i32, vectorised: *mut i32) -> i32;
    pub fn dp_model_in_flight(m: *const dp_model, in_flight: *mut usize) -> i32;
    pub fn dp_model_output_len(m: *const dp_model, noutput: *mut usize) -> i32;
    pub fn dp_host_cpu_budget() -> f64;
    pub fn dp_model_verifier_blob(m: *const dp_model, words: *mut *mut u64, nwords: *mut usize) -> i32;
    pub fn dp_verify_batch(ctx: *mut dp_ctx, verifier_blob: *const u64, blob_nwords: usize, proof_words: *const *const u64, proof_nwords: *const usize, inputs: *const i64, ninput: usize, outputs: *const i64, noutput: usize, nproofs: usize, threads: i32, results: *mut i32, wall_ms: *mut f64) -> i32;
    pub fn dp_verify(verifier_blob: *const u64, blob_nwords: usize, proof_words: *const u64, proof_nwords: usize, input: *const i64, ninput: usize, output: *const i64, noutput: usize) -> i32;
}

/// A failed call: the status code and the library's message for it (`dp_last_error`, thread local).
#[derive(Debug, Clone)]
pub struct DpError { pub code: i32, pub message: String }
impl core::fmt::Display for DpError {
    fn fmt(&self, f: &mut core::fmt::Formatter<'_>) -> core::fmt::Result { write!(f, "deep-prove-hip error {}: {}", self.code, self.message) }
}
impl std::error::Error for DpError {}
/// `Ok(())` for `DP_OK`, else the error with the library's message.
pub fn check(status: i32) -> Result<(), DpError> {
    if status == DP_OK { return Ok(()); }
    let message = unsafe { let p = dp_last_error(); if p.is_null() { String::new() } else { std::ffi::CStr::from_ptr(p).to_string_lossy().into_owned() } };
    Err(DpError { code: status, message })
}
/// A proof stream (or any buffer the library returned through `*mut *mut u64`): freed with `dp_free` on drop, never copied.
pub struct Words { ptr: *mut u64, len: usize }
impl Words {
    /// # Safety
    /// `ptr` / `len` come from one of the library's `proof_words` / `proof_nwords` out-parameters.
    pub unsafe fn from_raw(ptr: *mut u64, len: usize) -> Self { Self { ptr, len } }
    pub fn as_slice(&self) -> &[u64] { if self.ptr.is_null() { &[] } else { unsafe { core::slice::from_raw_parts(self.ptr, self.len) } } }
}
impl Drop for Words { fn drop(&mut self) { if !self.ptr.is_null() { unsafe { dp_free(self.ptr as *mut c_void) } } } }
// the buffer is plain memory owned by this value; the library's allocator is thread safe (csrc/capi.cpp, OutPool)
unsafe impl Send for Words {}
unsafe impl Sync for Words {}
