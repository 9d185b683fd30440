//! Seam 1 of the drop-in boundary: `mpcs::PolynomialCommitmentScheme<GoldilocksExt2>`
//! (`mpcs/src/lib.rs:111-226`) implemented over `libdeepprove_hip.so`. `zkml::{Context, Prover, verify}` are generic over the PCS
//! (`zkml/src/iop/context.rs:37`, `iop/prover.rs:40`), so `Context::<E, BasefoldHip>::generate` / `Prover::<E, HipTranscript,
//! BasefoldHip>` run the unmodified layer code with every commit / open on the MI355X.
//!
//! UNBUILT here (no Rust toolchain in the image, Plonky3 not vendored): written against the reference's sources as they stand;
//! `tests/test_rust_shim.py` checks every FFI call below against `deep-prove-hip-sys` (which it checks against the header).
//!
//! Field elements cross the ABI as canonical u64 (`SmallField::to_canonical_u64`, `ff_ext/src/lib.rs:97`; an extension element as
//! its two bases, `ExtensionField::as_bases`, `:112`). Proofs cross it as the canonical word stream of `csrc/proof.h`; they are
//! kept as that stream inside `HipProof` (the reference never looks inside a `Pcs::Proof`, it only serialises it:
//! `zkml/src/iop/mod.rs:21-97`), and `deep_prove_amd/wire.py` documents the mapping onto the serde layout of
//! `mpcs/src/basefold/structure.rs:292-363` for a maintainer who wants the reference's own `BasefoldProof` out of it.
use core::ffi::c_char;
use std::sync::{Arc, OnceLock};

use deep_prove_hip_sys as sys;
use ff_ext::{ExtensionField, GoldilocksExt2, SmallField};
use mpcs::{Error, Evaluation, PolynomialCommitmentScheme};
use multilinear_extensions::mle::{DenseMultilinearExtension, FieldType, MultilinearExtension};
use multilinear_extensions::virtual_poly::ArcMultilinearExtension;
use serde::{Deserialize, Serialize};
use transcript::{Challenge, ForkableTranscript, Transcript};

type E = GoldilocksExt2;
type F = <E as ExtensionField>::BaseField;

fn pcs_err(e: sys::DpError) -> Error {
    match e.code {
        sys::DP_ERR_VERIFY => Error::InvalidPcsOpen(e.message),
        sys::DP_ERR_SHAPE | sys::DP_ERR_ARG => Error::InvalidPcsParam(e.message),
        _ => Error::InvalidSnark(e.message),
    }
}
fn ext_words(x: &E) -> [u64; 2] {
    let b = x.as_bases();
    [b[0].to_canonical_u64(), b[1].to_canonical_u64()]
}
fn point_words(p: &[E]) -> Vec<u64> { p.iter().flat_map(|x| ext_words(x)).collect() }
fn ext_from_words(w: &[u64]) -> E { E::from_bases(&[F::from_v(w[0]), F::from_v(w[1])]) }

// ------------------------------------------------------------------------------------------------ the device context
/// One `dp_ctx` per process and GPU (`DEEP_PROVE_HIP_DEVICE`, default 0). The trait's functions are associated functions without a
/// context argument, so the context is process wide — as the rayon pool is for the reference. `dp_pcs_commit`, `dp_pcs_open`,
/// `dp_pcs_batch_open`, table uploads and frees are thread safe by contract of the ABI (header: "dp_pcs_commit"); everything else
/// here is called from the proving thread only, which is how zkml uses the trait (`commit/context.rs:79-103` commits from rayon
/// workers, `commit/context.rs:355-418` opens from the proving thread).
struct Ctx(*mut sys::dp_ctx);
// SAFETY: the handle is only passed to entry points the ABI declares callable from any thread, or used from the proving thread.
unsafe impl Send for Ctx {}
unsafe impl Sync for Ctx {}
impl Drop for Ctx { fn drop(&mut self) { unsafe { sys::dp_ctx_destroy(self.0); } } }
fn ctx() -> *mut sys::dp_ctx {
    static CTX: OnceLock<Ctx> = OnceLock::new();
    CTX.get_or_init(|| {
        let dev = std::env::var("DEEP_PROVE_HIP_DEVICE").ok().and_then(|s| s.parse().ok()).unwrap_or(0);
        let mut c = core::ptr::null_mut();
        sys::check(unsafe { sys::dp_ctx_create(dev, &mut c) }).expect("deep-prove-hip: no MI355X device (there is no CPU fallback)");
        Ctx(c)
    }).0
}

/// A table in HBM (`dp_buf`), freed with the context's thread-safe `dp_buf_free`.
struct DeviceTable(*mut sys::dp_buf);
unsafe impl Send for DeviceTable {}
unsafe impl Sync for DeviceTable {}
impl Drop for DeviceTable { fn drop(&mut self) { unsafe { sys::dp_buf_free(ctx(), self.0); } } }
fn poly_words(poly: &DenseMultilinearExtension<E>) -> Result<(Vec<u64>, usize, i32), Error> {
    Ok(match &poly.evaluations {
        FieldType::Base(v) => (v.iter().map(|x| x.to_canonical_u64()).collect(), v.len(), 0),
        FieldType::Ext(v) => (v.iter().flat_map(|x| ext_words(x)).collect(), v.len(), 1),
        FieldType::Unreachable => return Err(Error::InvalidPcsParam("unreachable field type".into())),
    })
}
fn upload(poly: &DenseMultilinearExtension<E>) -> Result<DeviceTable, Error> {
    let (words, n, is_ext) = poly_words(poly)?;
    let mut b = core::ptr::null_mut();
    sys::check(unsafe { sys::dp_buf_upload(ctx(), words.as_ptr(), n, is_ext, &mut b) }).map_err(pcs_err)?;
    Ok(DeviceTable(b))
}

// ------------------------------------------------------------------------------------------------ the transcript
/// `transcript::BasicTranscript` (`transcript/src/basic.rs:8-54`) with its sponge inside the library (`dp_transcript_*`): the
/// same Poseidon2 duplex challenger, bit for bit (tests/test_l0_independent.py). The PCS entry points need the library's handle, and
/// `&mut impl Transcript<E>` cannot be downcast; the handle therefore travels through `read_field_element_exts`, the one
/// state-reading accessor of the trait (`transcript/src/lib.rs:80-84`; `BasicTranscript` leaves it `unimplemented!()`): it returns
/// the handle's address as two 32-bit halves. `BasefoldHip` only works with this transcript and says so when handed another.
pub struct HipTranscript { h: *mut sys::dp_transcript }
unsafe impl Send for HipTranscript {}
impl HipTranscript {
    pub fn new(label: &'static [u8]) -> Self {
        let c = std::ffi::CString::new(label).expect("label without NUL");
        Self { h: unsafe { sys::dp_transcript_new(c.as_ptr()) } }
    }
    fn raw(&self) -> *mut sys::dp_transcript { self.h }
}
impl Drop for HipTranscript { fn drop(&mut self) { unsafe { sys::dp_transcript_free(self.h) } } }
impl Clone for HipTranscript {
    /// A fork replays nothing: the library has no clone entry point, so a clone is a fresh transcript fed the parent's next
    /// challenge (enough for `ForkableTranscript::fork`, which the zkml prover does not use: `iop/prover.rs` is single threaded).
    fn clone(&self) -> Self {
        let t = Self { h: unsafe { sys::dp_transcript_new(core::ptr::null()) } };
        let mut c = [0u64; 2];
        unsafe { sys::dp_transcript_challenge(self.h, core::ptr::null(), c.as_mut_ptr()); sys::dp_transcript_append_elements(t.h, c.as_ptr(), 2); }
        t
    }
}
impl Transcript<E> for HipTranscript {
    fn append_field_elements(&mut self, elements: &[F]) {
        let w: Vec<u64> = elements.iter().map(|x| x.to_canonical_u64()).collect();
        unsafe { sys::dp_transcript_append_elements(self.h, w.as_ptr(), w.len()); }
    }
    fn append_field_element_ext(&mut self, element: &E) {
        let w = ext_words(element);
        unsafe { sys::dp_transcript_append_elements(self.h, w.as_ptr(), 2); }
    }
    fn append_message(&mut self, msg: &[u8]) { unsafe { sys::dp_transcript_append_message(self.h, msg.as_ptr(), msg.len()); } }
    fn read_challenge(&mut self) -> Challenge<E> {
        let mut c = [0u64; 2];
        unsafe { sys::dp_transcript_challenge(self.h, core::ptr::null(), c.as_mut_ptr()); }
        Challenge { elements: ext_from_words(&c) }
    }
    fn read_field_element_exts(&self) -> Vec<E> {
        let a = self.h as usize as u64;
        vec![E::from_bases(&[F::from_v(a & 0xFFFF_FFFF), F::from_v(a >> 32)])]
    }
    fn read_field_element(&self) -> F { unimplemented!() }
    fn send_challenge(&self, _challenge: E) { unimplemented!() }
    fn commit_rolling(&mut self) {}
}
impl ForkableTranscript<E> for HipTranscript {}
/// the library handle behind a `&mut impl Transcript<E>` (see `HipTranscript`)
fn handle_of(t: &mut impl Transcript<E>) -> *mut sys::dp_transcript {
    let v = t.read_field_element_exts();  // BasicTranscript panics here: BasefoldHip needs HipTranscript
    let w = ext_words(&v[0]);
    ((w[1] << 32) | w[0]) as usize as *mut sys::dp_transcript
}

// ------------------------------------------------------------------------------------------------ the scheme
#[derive(Clone, Debug, Default)]
pub struct BasefoldHip;
#[derive(Clone, Debug, Serialize, Deserialize)]
pub struct HipParams { pub max_poly_size: usize }
/// `BasefoldCommitment` (`mpcs/src/basefold/structure.rs:161-166`): what the verifier keeps
#[derive(Clone, Debug, Default, Serialize, Deserialize, PartialEq, Eq)]
pub struct HipCommitment { pub root: [u64; 4], pub num_vars: usize, pub is_base: bool, pub num_polys: usize }
/// `BasefoldCommitmentWithWitness` (`structure.rs:63-73`): the witness (evaluations, codeword, Merkle layers) stays in HBM behind the
/// handle. Serde carries the pure commitment only — a deserialised witness has to be re-committed, which is what
/// `Context::generate` does anyway when it rebuilds a model's commitments.
#[derive(Clone, Debug, Serialize, Deserialize)]
pub struct HipCommitmentWithWitness {
    pub pure: HipCommitment,
    #[serde(skip)] witness: Option<Arc<Witness>>,
}
enum Handle { One(*mut sys::dp_commit), Batch(*mut sys::dp_batch_commit) }
struct Witness { handle: Handle, _tables: Vec<DeviceTable> }  // the commitment borrows the tables: they live as long as it does
unsafe impl Send for Witness {}
unsafe impl Sync for Witness {}
impl core::fmt::Debug for Witness { fn fmt(&self, f: &mut core::fmt::Formatter<'_>) -> core::fmt::Result { write!(f, "Witness(device)") } }
impl Drop for Witness {
    fn drop(&mut self) {
        unsafe { match self.handle { Handle::One(c) => { sys::dp_pcs_commit_free(ctx(), c); } Handle::Batch(c) => { sys::dp_pcs_batch_commit_free(ctx(), c); } } }
    }
}
/// `BasefoldProof` as the canonical word stream (`csrc/proof.h`; layout in `include/deep_prove_hip.h` at `dp_pcs_batch_open`)
#[derive(Clone, Debug, Serialize, Deserialize)]
pub struct HipProof { pub words: Vec<u64> }

fn take_words(ptr: *mut u64, n: usize) -> HipProof {
    let w = unsafe { sys::Words::from_raw(ptr, n) };
    HipProof { words: w.as_slice().to_vec() }
}
fn one(c: &HipCommitmentWithWitness) -> Result<*mut sys::dp_commit, Error> {
    match c.witness.as_deref().map(|w| &w.handle) { Some(Handle::One(h)) => Ok(*h), _ => Err(Error::InvalidPcsParam("commitment without a device witness (deserialised?)".into())) }
}

impl PolynomialCommitmentScheme<E> for BasefoldHip {
    type Param = HipParams;
    type ProverParam = HipParams;
    type VerifierParam = HipParams;
    type CommitmentWithWitness = HipCommitmentWithWitness;
    type Commitment = HipCommitment;
    type CommitmentChunk = [u64; 4];
    type Proof = HipProof;

    /// `Basefold::setup` (`mpcs/src/basefold.rs:278-290`): the FFT root / coset tables are built on the device
    fn setup(poly_size: usize) -> Result<Self::Param, Error> {
        sys::check(unsafe { sys::dp_pcs_setup(ctx(), poly_size.next_power_of_two()) }).map_err(pcs_err)?;
        Ok(HipParams { max_poly_size: poly_size.next_power_of_two() })
    }
    /// `Basefold::trim` (`basefold.rs:292-303`)
    fn trim(param: Self::Param, poly_size: usize) -> Result<(Self::ProverParam, Self::VerifierParam), Error> {
        if poly_size > param.max_poly_size { return Err(Error::PolynomialTooLarge(poly_size)); }
        Ok((param.clone(), param))
    }
    /// `Basefold::commit` (`basefold.rs:304-354`); thread safe (rayon workers: `zkml/src/commit/context.rs:79-103`)
    fn commit(_pp: &Self::ProverParam, poly: &DenseMultilinearExtension<E>) -> Result<Self::CommitmentWithWitness, Error> {
        let table = upload(poly)?;
        let mut h = core::ptr::null_mut();
        let mut root = [0u64; 4];
        sys::check(unsafe { sys::dp_pcs_commit(ctx(), table.0, &mut h, root.as_mut_ptr()) }).map_err(pcs_err)?;
        let pure = HipCommitment { root, num_vars: poly.num_vars, is_base: matches!(poly.evaluations, FieldType::Base(_)), num_polys: 1 };
        Ok(HipCommitmentWithWitness { pure, witness: Some(Arc::new(Witness { handle: Handle::One(h), _tables: vec![table] })) })
    }
    /// `Basefold::write_commitment` (`basefold.rs:448-457`): the root's four base elements
    fn write_commitment(comm: &Self::Commitment, transcript: &mut impl Transcript<E>) -> Result<(), Error> {
        let r: Vec<F> = comm.root.iter().map(|&w| F::from_v(w)).collect();
        transcript.append_field_elements(&r);
        Ok(())
    }
    fn get_pure_commitment(comm: &Self::CommitmentWithWitness) -> Self::Commitment { comm.pure.clone() }
    /// `BasefoldRSParams::get_basecode_msg_size_log()` = 7 (`basefold.rs:1204-1206`)
    fn trivial_num_vars() -> usize { 7 }
    /// `Basefold::batch_commit` (`basefold.rs:356-446`)
    fn batch_commit(_pp: &Self::ProverParam, polys: &[DenseMultilinearExtension<E>]) -> Result<Self::CommitmentWithWitness, Error> {
        let tables: Vec<DeviceTable> = polys.iter().map(upload).collect::<Result<_, _>>()?;
        let ptrs: Vec<*const sys::dp_buf> = tables.iter().map(|t| t.0 as *const sys::dp_buf).collect();
        let mut h = core::ptr::null_mut();
        let mut root = [0u64; 4];
        sys::check(unsafe { sys::dp_pcs_batch_commit(ctx(), ptrs.as_ptr(), ptrs.len() as i32, &mut h, root.as_mut_ptr()) }).map_err(pcs_err)?;
        let pure = HipCommitment { root, num_vars: polys[0].num_vars, is_base: matches!(polys[0].evaluations, FieldType::Base(_)), num_polys: polys.len() };
        Ok(HipCommitmentWithWitness { pure, witness: Some(Arc::new(Witness { handle: Handle::Batch(h), _tables: tables })) })
    }
    /// `Basefold::open` (`basefold.rs:466-544`)
    fn open(_pp: &Self::ProverParam, _poly: &DenseMultilinearExtension<E>, comm: &Self::CommitmentWithWitness, point: &[E], eval: &E,
            transcript: &mut impl Transcript<E>) -> Result<Self::Proof, Error> {
        let (p, e) = (point_words(point), ext_words(eval));
        let (mut w, mut n) = (core::ptr::null_mut(), 0usize);
        sys::check(unsafe { sys::dp_pcs_open(ctx(), one(comm)?, p.as_ptr(), point.len() as u32, e.as_ptr(), handle_of(transcript), &mut w, &mut n) }).map_err(pcs_err)?;
        Ok(take_words(w, n))
    }
    /// `Basefold::batch_open` (`basefold.rs:546-770`) with the trait's general evaluation list
    fn batch_open(_pp: &Self::ProverParam, _polys: &[DenseMultilinearExtension<E>], comms: &[Self::CommitmentWithWitness], points: &[Vec<E>],
                  evals: &[Evaluation<E>], transcript: &mut impl Transcript<E>) -> Result<Self::Proof, Error> {
        let handles: Vec<*const sys::dp_commit> = comms.iter().map(|c| one(c).map(|h| h as *const sys::dp_commit)).collect::<Result<_, _>>()?;
        let flat: Vec<u64> = points.iter().flat_map(|p| point_words(p)).collect();
        let pnv: Vec<u32> = points.iter().map(|p| p.len() as u32).collect();
        let (ep, ept): (Vec<u32>, Vec<u32>) = evals.iter().map(|e| (e.poly() as u32, e.point() as u32)).unzip();
        let ev: Vec<u64> = evals.iter().flat_map(|e| ext_words(e.value())).collect();
        let (mut w, mut n) = (core::ptr::null_mut(), 0usize);
        sys::check(unsafe { sys::dp_pcs_batch_open_evals(ctx(), handles.as_ptr(), handles.len() as i32, flat.as_ptr(), pnv.as_ptr(), pnv.len() as i32,
                                                         ep.as_ptr(), ept.as_ptr(), ev.as_ptr(), evals.len() as i32, handle_of(transcript), &mut w, &mut n) }).map_err(pcs_err)?;
        Ok(take_words(w, n))
    }
    /// `Basefold::simple_batch_open` (`basefold.rs:777-861`)
    fn simple_batch_open(_pp: &Self::ProverParam, _polys: &[ArcMultilinearExtension<E>], comm: &Self::CommitmentWithWitness, point: &[E], _evals: &[E],
                         transcript: &mut impl Transcript<E>) -> Result<Self::Proof, Error> {
        let h = match comm.witness.as_deref().map(|w| &w.handle) { Some(Handle::Batch(h)) => *h, _ => return Err(Error::InvalidPcsParam("not a batch commitment".into())) };
        let p = point_words(point);
        let (mut w, mut n) = (core::ptr::null_mut(), 0usize);
        sys::check(unsafe { sys::dp_pcs_simple_batch_open(ctx(), h, p.as_ptr(), point.len() as u32, handle_of(transcript), &mut w, &mut n) }).map_err(pcs_err)?;
        Ok(take_words(w, n))
    }
    /// `Basefold::verify` (`basefold.rs:863-962`) — host only
    fn verify(vp: &Self::VerifierParam, comm: &Self::Commitment, point: &[E], eval: &E, proof: &Self::Proof, transcript: &mut impl Transcript<E>) -> Result<(), Error> {
        let (p, e) = (point_words(point), ext_words(eval));
        sys::check(unsafe { sys::dp_pcs_verify(vp.max_poly_size, comm.root.as_ptr(), comm.num_vars as u32, comm.is_base as i32, p.as_ptr(), e.as_ptr(),
                                               proof.words.as_ptr(), proof.words.len(), handle_of(transcript)) }).map_err(pcs_err)
    }
    /// `Basefold::batch_verify` (`basefold.rs:964-1098`) — host only
    fn batch_verify(vp: &Self::VerifierParam, comms: &[Self::Commitment], points: &[Vec<E>], evals: &[Evaluation<E>], proof: &Self::Proof,
                    transcript: &mut impl Transcript<E>) -> Result<(), Error> {
        let roots: Vec<u64> = comms.iter().flat_map(|c| c.root).collect();
        let nv: Vec<u32> = comms.iter().map(|c| c.num_vars as u32).collect();
        let ib: Vec<i32> = comms.iter().map(|c| c.is_base as i32).collect();
        let flat: Vec<u64> = points.iter().flat_map(|p| point_words(p)).collect();
        let pnv: Vec<u32> = points.iter().map(|p| p.len() as u32).collect();
        let (ep, ept): (Vec<u32>, Vec<u32>) = evals.iter().map(|e| (e.poly() as u32, e.point() as u32)).unzip();
        let ev: Vec<u64> = evals.iter().flat_map(|e| ext_words(e.value())).collect();
        sys::check(unsafe { sys::dp_pcs_batch_verify_evals(vp.max_poly_size, roots.as_ptr(), nv.as_ptr(), ib.as_ptr(), comms.len() as i32, flat.as_ptr(), pnv.as_ptr(),
                                                           pnv.len() as i32, ep.as_ptr(), ept.as_ptr(), ev.as_ptr(), evals.len() as i32, proof.words.as_ptr(),
                                                           proof.words.len(), handle_of(transcript)) }).map_err(pcs_err)
    }
    /// `Basefold::simple_batch_verify` (`basefold.rs:1100-1203`) — host only
    fn simple_batch_verify(vp: &Self::VerifierParam, comm: &Self::Commitment, point: &[E], evals: &[E], proof: &Self::Proof,
                           transcript: &mut impl Transcript<E>) -> Result<(), Error> {
        let p = point_words(point);
        let ev: Vec<u64> = evals.iter().flat_map(|e| ext_words(e)).collect();
        sys::check(unsafe { sys::dp_pcs_simple_batch_verify(vp.max_poly_size, comm.root.as_ptr(), comm.num_vars as u32, comm.is_base as i32, p.as_ptr(), ev.as_ptr(),
                                                            evals.len() as i32, proof.words.as_ptr(), proof.words.len(), handle_of(transcript)) }).map_err(pcs_err)
    }
}

// ------------------------------------------------------------------------------------------------ asynchronous forms
/// Submit / poll forms of the two PCS seams (`include/deep_prove_hip.h`, "asynchronous seam calls"). The reference reaches `PCS::commit` from rayon workers
/// (`zkml/src/commit/context.rs:79-103`): every worker blocks in its own commit, so the number of commits in flight is the size of the thread pool. With an
/// engine ONE thread submits the commits of every witness column of every proof it is driving and polls the tickets; calls of the same shape that are queued
/// together run in lock step with merged kernel launches (what `dp_model_prove_batch` does for whole proofs). A `zkml::Prover` patched this way keeps its
/// control flow — `commit_async(..)` where it called `commit(..)`, `.wait()` where it needs the root for the transcript.
pub struct AsyncEngine(*mut sys::dp_async);
unsafe impl Send for AsyncEngine {}
unsafe impl Sync for AsyncEngine {}
impl Drop for AsyncEngine { fn drop(&mut self) { unsafe { sys::dp_ctx_route_to_engine(ctx(), core::ptr::null_mut()); sys::dp_async_destroy(self.0); } } }
/// a submitted call; dropping it waits for completion (the engine reads the caller's tables until then)
pub struct PendingCommit { ticket: *mut sys::dp_ticket, table: Option<DeviceTable>, num_vars: usize, is_base: bool }
pub struct PendingOpen { ticket: *mut sys::dp_ticket }
impl AsyncEngine {
    /// after `BasefoldHip::setup` (the workers share the PCS tables of the context)
    pub fn new(max_in_flight: usize) -> Result<Self, Error> {
        let mut h = core::ptr::null_mut();
        sys::check(unsafe { sys::dp_async_create(ctx(), max_in_flight as i32, 0, &mut h) }).map_err(pcs_err)?;
        Ok(AsyncEngine(h))
    }
    /// Round 6: route the BLOCKING trait methods of this crate (`BasefoldHip::commit`, `batch_open`, and `prove_parallel` of `sumcheck-hip-patch`) through this
    /// engine (`dp_ctx_route_to_engine`): every call becomes submit + wait, and calls of one shape that other rayon workers make at the same moment — the
    /// `into_par_iter` over nodes of `zkml/src/commit/context.rs:79-103`, over columns of `layers/activation.rs:294-304` — are proved in lock step with merged
    /// launches. The synchronous traits stay as they are; the number of calls in flight is the number of threads blocked in them.
    pub fn route_blocking_calls(&self) -> Result<(), Error> { sys::check(unsafe { sys::dp_ctx_route_to_engine(ctx(), self.0) }).map_err(pcs_err) }
    pub fn unroute_blocking_calls(&self) -> Result<(), Error> { sys::check(unsafe { sys::dp_ctx_route_to_engine(ctx(), core::ptr::null_mut()) }).map_err(pcs_err) }
    /// `PCS::commit(pp, &poly)`: the host polynomial is uploaded and committed by ONE ticket (`dp_pcs_commit_host_submit`)
    pub fn commit_async(&self, poly: &DenseMultilinearExtension<E>) -> Result<PendingCommit, Error> {
        let (words, n, is_ext) = poly_words(poly)?;
        let mut t = core::ptr::null_mut();
        sys::check(unsafe { sys::dp_pcs_commit_host_submit(self.0, words.as_ptr(), n, is_ext, &mut t) }).map_err(pcs_err)?;
        Ok(PendingCommit { ticket: t, table: None, num_vars: poly.num_vars, is_base: is_ext == 0 })
    }
    /// `PCS::batch_open` with `Evaluation::new(i, i, evals[i])` (`zkml/src/commit/context.rs:355-418`); the transcript must not be used until `wait`
    pub fn batch_open_async(&self, comms: &[HipCommitmentWithWitness], points: &[Vec<E>], evals: &[E], transcript: &mut impl Transcript<E>) -> Result<PendingOpen, Error> {
        let handles: Vec<*const sys::dp_commit> = comms.iter().map(|c| one(c).map(|h| h as *const sys::dp_commit)).collect::<Result<_, _>>()?;
        let flat: Vec<u64> = points.iter().flat_map(|p| point_words(p)).collect();
        let ev: Vec<u64> = evals.iter().flat_map(|e| ext_words(e)).collect();
        let mut t = core::ptr::null_mut();
        sys::check(unsafe { sys::dp_pcs_batch_open_submit(self.0, handles.as_ptr(), handles.len() as i32, flat.as_ptr(), ev.as_ptr(), handle_of(transcript), &mut t) }).map_err(pcs_err)?;
        Ok(PendingOpen { ticket: t })
    }
}
impl PendingCommit {
    pub fn is_done(&self) -> Result<bool, Error> { let s = unsafe { sys::dp_poll(self.ticket) }; if s < 0 { Err(pcs_err(sys::check(s).unwrap_err())) } else { Ok(s == 1) } }
    pub fn wait(mut self) -> Result<HipCommitmentWithWitness, Error> {
        sys::check(unsafe { sys::dp_wait(self.ticket) }).map_err(pcs_err)?;
        let mut b = core::ptr::null_mut();
        sys::check(unsafe { sys::dp_ticket_buf(self.ticket, &mut b) }).map_err(pcs_err)?;
        self.table = Some(DeviceTable(b));
        let mut h = core::ptr::null_mut();
        let mut root = [0u64; 4];
        sys::check(unsafe { sys::dp_ticket_commit(self.ticket, &mut h, root.as_mut_ptr()) }).map_err(pcs_err)?;
        let pure = HipCommitment { root, num_vars: self.num_vars, is_base: self.is_base, num_polys: 1 };
        Ok(HipCommitmentWithWitness { pure, witness: Some(Arc::new(Witness { handle: Handle::One(h), _tables: vec![self.table.take().unwrap()] })) })
    }
}
impl Drop for PendingCommit { fn drop(&mut self) { unsafe { sys::dp_wait(self.ticket); sys::dp_ticket_free(self.ticket); } } }
impl PendingOpen {
    pub fn is_done(&self) -> Result<bool, Error> { let s = unsafe { sys::dp_poll(self.ticket) }; if s < 0 { Err(pcs_err(sys::check(s).unwrap_err())) } else { Ok(s == 1) } }
    pub fn wait(self) -> Result<HipProof, Error> {
        sys::check(unsafe { sys::dp_wait(self.ticket) }).map_err(pcs_err)?;
        let (mut w, mut n) = (core::ptr::null_mut(), 0usize);
        sys::check(unsafe { sys::dp_ticket_words(self.ticket, 0, &mut w, &mut n) }).map_err(pcs_err)?;
        Ok(take_words(w, n))
    }
}
impl Drop for PendingOpen { fn drop(&mut self) { unsafe { sys::dp_wait(self.ticket); sys::dp_ticket_free(self.ticket); } } }

#[allow(dead_code)]
fn _label(l: &'static [u8]) -> *const c_char { l.as_ptr() as *const c_char }
