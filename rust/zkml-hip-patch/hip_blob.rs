//! Seam 3 (INTEGRATION.md "model blob"): the reference's padded, quantised `Model<Element>` written as the int64 blob
//! `dp_model_setup` reads, and `Prover::prove` for a whole batch of inputs through `dp_model_prove_batch` — the path the headline
//! throughput is measured on (cohorts, device-side Fiat-Shamir, fused tails all live behind it). Add to the `zkml` crate as
//! `zkml/src/hip_blob.rs` with `mod hip_blob;` in `lib.rs` and a dependency on `deep-prove-hip-sys`: the module reads `pub(crate)` fields
//! of the layer structs; where a field is private to its module the accessor named in the comment has to be added next to the struct
//! (each is a one-line getter; `accessors.patch` beside this file lists every one of them as a hunk against the reference). UNBUILT in the repository's environment (no Rust toolchain); what IS checked on every CPU test run
//! (`tests/test_rust_shim.py`): the kind numbers below against `enum LayerKind` of `csrc/proof.h`, that every variant of the reference's
//! `Layer` enum (`zkml/src/layers/mod.rs:66-93`) has an arm here, and every `sys::dp_*` call against the extern block. The word order per
//! kind follows `csrc/blob.h` (the parser) and `deep-prove_amd/models.py` (the writer the golden fixtures come from).
//!
//! Blob (include/deep_prove_hip.h "model blob", GRAPH form): input_len, -(#nodes), #inputs, len.., #outputs, (node, slot).., then per node
//! kind, #in, (node, slot).., parameters. Node ids are renumbered to positions in `to_forward_iterator()` order (a node only reads earlier ones).
use std::collections::HashMap;

use deep_prove_hip_sys as sys;

use crate::{
    Element,
    layers::{Layer, activation::Activation, matrix_mul::OperandMatrix, pooling::Pooling, provable::{Edge, NodeId}, transformer::positional::Positional},
    model::Model,
    tensor::Tensor,
};

pub const KIND_DENSE: i64 = 0;
pub const KIND_REQUANT: i64 = 1;
pub const KIND_RELU: i64 = 2;
pub const KIND_CONV: i64 = 3;
pub const KIND_MAXPOOL: i64 = 4;
pub const KIND_FLATTEN: i64 = 5;
pub const KIND_MATMUL: i64 = 6;
pub const KIND_ADD: i64 = 7;
pub const KIND_EMBED: i64 = 8;
pub const KIND_POSITIONAL: i64 = 9;
pub const KIND_MATMUL2: i64 = 10;
pub const KIND_ADD2: i64 = 11;
pub const KIND_CONCAT_MATMUL: i64 = 12;
pub const KIND_QKV: i64 = 13;
pub const KIND_LAYERNORM: i64 = 14;
pub const KIND_SOFTMAX: i64 = 15;
pub const KIND_MHA: i64 = 16;
pub const KIND_GELU: i64 = 17;

#[derive(Debug)]
pub enum BlobError { Unsupported(String), Shape(String), Ffi(sys::DpError) }
impl From<sys::DpError> for BlobError { fn from(e: sys::DpError) -> Self { BlobError::Ffi(e) } }

fn data(t: &Tensor<Element>) -> impl Iterator<Item = i64> + '_ { t.get_data().iter().map(|&v| v as i64) }
fn dims3(s: &[usize]) -> Result<[i64; 3], BlobError> {
    if s.len() != 3 { return Err(BlobError::Shape(format!("rank-3 shape expected, found {s:?}"))); }
    Ok([s[0] as i64, s[1] as i64, s[2] as i64])
}

/// The parameters of a softmax as kinds 15 / 16 list them after the shape: multiplier, 1 / temperature bits, input scale bits, table size,
/// bkm, zero chunks, zero table vars, allowable error. Needs `Softmax::quant_info()` -> `&QuantisedSoftmaxData` and getters on it
/// (`softmax.rs:82-97`: all fields are private to the module): multiplier = `(SCALE_FACTOR as f32 * input_scale_factor.scale()).round()`,
/// table size = `lut.size()`, allowable error = `(error_bound * OUTPUT_SCALE_FACTOR as f32).round()` (`SoftmaxCtx`, softmax.rs:1153-1169).
fn softmax_words(s: &crate::layers::transformer::softmax::Softmax<Element>) -> Result<[i64; 8], BlobError> {
    let q = s.quant_info().ok_or_else(|| BlobError::Unsupported("softmax without quantisation data".into()))?;
    Ok([q.multiplier() as i64, q.inv_float_temperature().to_bits() as i64, q.input_scale_factor().scale().to_bits() as i64, q.table_size() as i64,
        q.bkm() as i64, q.number_zero_chunks() as i64, q.zero_table_vars() as i64, q.allowable_error() as i64])
}

/// `Model<Element>` (after `pad_model` and quantisation: every tensor padded to powers of two) -> the blob. `input_shapes`: the PADDED input
/// shapes (`Model::input_shapes`, model/mod.rs:104-106).
pub fn model_to_blob(model: &Model<Element>) -> Result<Vec<i64>, BlobError> {
    let order: Vec<(NodeId, &crate::layers::provable::Node<Element>)> = model.to_forward_iterator().map(|(id, n)| (*id, n)).collect();
    let pos: HashMap<NodeId, i64> = order.iter().enumerate().map(|(i, (id, _))| (*id, i as i64)).collect();
    let edge = |e: &Edge| -> [i64; 2] { [e.node.map(|n| pos[&n]).unwrap_or(-1), e.index as i64] };
    let input_lens: Vec<i64> = model.input_shapes().iter().map(|s| s.product() as i64).collect();
    let mut w: Vec<i64> = vec![input_lens.iter().sum(), -(order.len() as i64), input_lens.len() as i64];
    w.extend(&input_lens);
    // output tensors: the wires whose reader is the model (OutputWire with node None), in the order of the output index
    let mut outs: Vec<(usize, [i64; 2])> = Vec::new();
    for (i, (_, n)) in order.iter().enumerate() {
        for (slot, wire) in n.outputs.iter().enumerate() {
            for e in wire.edges.iter().filter(|e| e.node.is_none()) { outs.push((e.index, [i as i64, slot as i64])); }
        }
    }
    outs.sort();
    w.push(outs.len() as i64);
    for (_, o) in &outs { w.extend(o); }
    for (_, node) in &order {
        let head = |w: &mut Vec<i64>, kind: i64| { w.push(kind); w.push(node.inputs.len() as i64); for e in &node.inputs { w.extend(edge(e)); } };
        match &node.operation {
            Layer::Dense(d) => {  // [0, rows, cols, matrix row major, bias]
                let s = d.matrix.get_shape();
                head(&mut w, KIND_DENSE); w.extend([s[0] as i64, s[1] as i64]); w.extend(data(&d.matrix)); w.extend(data(&d.bias));
            }
            Layer::Requant(r) => { head(&mut w, KIND_REQUANT); w.extend([r.right_shift as i64, r.fp_scale as i64, r.fixed_point_multiplier as i64, r.intermediate_bit_size as i64]); }
            // Activation::Gelu: one word, GELUQuantData::multiplier (layers/activation.rs:565-572; the field is private there: this patch adds
            // `pub(crate) fn multiplier(&self) -> Element` next to table_size() and `pub(crate) fn quant_data(&self) -> Option<&GELUQuantData>` on GELU). The library opens the scaled column at the claim verify_activation
            // files (:495-505), not at the divided one the reference's prove_step files (:405-430), so its proofs verify at every size.
            Layer::Activation(Activation::Relu(_)) => head(&mut w, KIND_RELU),
            Layer::Activation(Activation::Gelu(g)) => { head(&mut w, KIND_GELU); w.push(g.quant_data().ok_or_else(|| BlobError::Unsupported("GELU not quantized".into()))?.multiplier() as i64); }
            Layer::Flatten(_) | Layer::Reshape(_) => head(&mut w, KIND_FLATTEN),  // tensors cross the ABI flat: the claim passes through
            Layer::Convolution(c) => {  // [3, kw, kx, kernel side, input side, unpadded output shape (3), filter, bias]: needs the UN-FFTed padded filter,
                // `Convolution::padded_filter()` (the quantised op keeps the FFT, convolution.rs:52-60; the library transforms the kernels itself)
                let f = c.padded_filter(); let s = f.get_shape(); let nw = c.padded_input_side(); let up = c.unpadded_output_shape();
                head(&mut w, KIND_CONV); w.extend([s[0] as i64, s[1] as i64, s[2] as i64, nw as i64, up[0] as i64, up[1] as i64, up[2] as i64]); w.extend(data(&f)); w.extend(data(&c.bias));
            }
            Layer::Pooling(Pooling::Maxpool2D(p)) => {  // [4, padded input shape (3)]: kernel 2, stride 2 only
                if p.kernel_size != 2 || p.stride != 2 { return Err(BlobError::Unsupported("maxpool other than 2x2 stride 2".into())); }
                let s = dims3(&model.padded_input_shape_of(node)?)?; head(&mut w, KIND_MAXPOOL); w.extend(s);
            }
            Layer::MatMul(m) => match (&m.left_matrix, &m.right_matrix) {
                (OperandMatrix::Input, OperandMatrix::Weight(r)) => {  // [6, k, n, flags (1 bias, 2 TransposeB), matrix, bias]
                    let s = r.tensor.get_shape(); let t = m.is_transposed_b();
                    let (k, n) = if t { (s[1], s[0]) } else { (s[0], s[1]) };
                    head(&mut w, KIND_MATMUL); w.extend([k as i64, n as i64, (m.bias.is_some() as i64) | ((t as i64) << 1)]); w.extend(data(&r.tensor));
                    if let Some(b) = &m.bias { w.extend(data(b)); }
                }
                (OperandMatrix::Input, OperandMatrix::Input) => {  // [10, k, n, flags (2 TransposeB)]
                    let (k, n) = m.inner_and_output_dims(&model.padded_input_shapes_of(node)?)?;
                    head(&mut w, KIND_MATMUL2); w.extend([k as i64, n as i64, (m.is_transposed_b() as i64) << 1]);
                }
                _ => return Err(BlobError::Unsupported("MatMul with a constant LEFT matrix".into())),
            },
            Layer::Add(a) => match a.operand() {  // `Add::operand()` / `Add::multipliers()`: getters for the private fields (add.rs:37-42, 285-294)
                Some((t, _)) => { let (l, r) = a.multipliers(); head(&mut w, KIND_ADD); w.extend([l as i64, r as i64, t.get_data().len() as i64]); w.extend(data(t)); }
                None => { let (l, r) = a.multipliers(); head(&mut w, KIND_ADD2); w.extend([l as i64, r as i64]); }
            },
            Layer::Embeddings(e) => { let t = e.table(); let s = t.get_shape(); head(&mut w, KIND_EMBED); w.extend([s[0] as i64, s[1] as i64]); w.extend(data(t)); }
            Layer::Positional(Positional::Learned(p)) => {  // [9, left, right, positions, embedding size, table]
                let s = p.positional.get_shape(); let (l, r) = p.multipliers();
                head(&mut w, KIND_POSITIONAL); w.extend([l as i64, r as i64, s[0] as i64, s[1] as i64]); w.extend(data(&p.positional));
            }
            Layer::QKV(q) => {  // [13, k, n, W_q | W_k | W_v, b_q | b_k | b_v]
                let s = q.q.get_shape();
                head(&mut w, KIND_QKV); w.extend([s[0] as i64, s[1] as i64]);
                for t in [&q.q, &q.k, &q.v, &q.q_bias, &q.k_bias, &q.v_bias] { w.extend(data(t)); }
            }
            Layer::ConcatMatMul(c) => {  // [12, shape A, shape B, (concat, mat_mul, output) of A, of B, 0 | 1 + permutation]; `ConcatMatMul::permutations()`
                let shapes = model.padded_input_shapes_of(node)?; let p = c.permutations();
                head(&mut w, KIND_CONCAT_MATMUL); w.extend(dims3(&shapes[0])?); w.extend(dims3(&shapes[1])?);
                w.extend(p.left.as_array().map(|x| x as i64)); w.extend(p.right.as_array().map(|x| x as i64));
                match &p.permute { Some(pm) => { w.push(1); w.extend(pm.as_slice().iter().map(|&x| x as i64)); } None => w.push(0) }
            }
            Layer::LayerNorm(l) => {  // [14, dim, N, multiplier, eps bits, range check bits, log2 top chunk scalar, gamma, beta]
                let q = l.quant_info.as_ref().ok_or_else(|| BlobError::Unsupported("layernorm without quantisation data".into()))?;
                head(&mut w, KIND_LAYERNORM);
                w.extend([l.gamma.get_data().len() as i64, q.dim_size() as i64, q.multiplier() as i64, q.lut_eps_bits() as i64, q.range_check_bits() as i64, q.top_chunk_scalar_log() as i64]);
                w.extend(data(&l.gamma)); w.extend(data(&l.beta));
            }
            Layer::Softmax(s) => { let sh = dims3(&model.padded_input_shape_of(node)?)?; head(&mut w, KIND_SOFTMAX); w.extend(sh); w.extend(softmax_words(s)?); }
            Layer::Mha(m) => {  // [16, context length, heads, head_dim (padded), the softmax's parameters]; `Mha::softmax()`, `padded_dims()`
                let (seq, heads, head_dim) = m.padded_dims(&model.padded_input_shapes_of(node)?)?;
                head(&mut w, KIND_MHA); w.extend([seq as i64, heads as i64, head_dim as i64]); w.extend(softmax_words(m.softmax())?);
            }
            Layer::SchoolBookConvolution(_) => return Err(BlobError::Unsupported("SchoolBookConvolution has no proof in the reference either".into())),
            Layer::Logits(_) => return Err(BlobError::Unsupported("Logits: prover and verifier transcripts disagree outside cfg(test) (iop/prover.rs:423-435 vs logits.rs:680-690)".into())),
        }
    }
    Ok(w)
}

/// A model set up on one GPU: `Context::generate` (model commitments, table commitments) happened inside `dp_model_setup`.
pub struct HipModel { ctx: *mut sys::dp_ctx, model: *mut sys::dp_model, input_len: usize, output_cap: usize }
// the library serialises what has to be serialised (per-proof arenas and streams; the model commitments are read-only)
unsafe impl Send for HipModel {}
impl Drop for HipModel { fn drop(&mut self) { unsafe { sys::dp_model_free(self.model); sys::dp_ctx_destroy(self.ctx); } } }

impl HipModel {
    pub fn setup(device_id: i32, model: &Model<Element>, output_cap: usize) -> Result<Self, BlobError> {
        let blob = model_to_blob(model)?;
        let (mut ctx, mut m) = (core::ptr::null_mut(), core::ptr::null_mut());
        sys::check(unsafe { sys::dp_ctx_create(device_id, &mut ctx) })?;
        if let Err(e) = sys::check(unsafe { sys::dp_model_setup(ctx, blob.as_ptr(), blob.len(), &mut m) }) { unsafe { sys::dp_ctx_destroy(ctx); } return Err(e.into()); }
        Ok(Self { ctx, model: m, input_len: blob[0] as usize, output_cap })
    }

    /// `Prover::prove` for every input (each the concatenation of the model's padded, quantised input tensors) with up to `in_flight` proofs on
    /// the GPU at once: (canonical proof stream, model output) per input. Streams go to `zkml::Proof` through the wire format of
    /// deep-prove_amd/wire.py (`to_rmp`), or to `dp_verify` as they are.
    pub fn prove_batch(&self, inputs: &[Vec<Element>], in_flight: i32) -> Result<Vec<(sys::Words, Vec<Element>)>, BlobError> {
        let n = inputs.len();
        let mut flat: Vec<i64> = Vec::with_capacity(n * self.input_len);
        for x in inputs {
            if x.len() != self.input_len { return Err(BlobError::Shape(format!("input of {} words, the model takes {}", x.len(), self.input_len))); }
            flat.extend(x.iter().map(|&v| v as i64));
        }
        let mut proofs: Vec<*mut u64> = vec![core::ptr::null_mut(); n];
        let mut lens = vec![0usize; n];
        let mut outs = vec![0i64; n * self.output_cap];
        let (mut nout, mut wall) = (0usize, 0f64);
        sys::check(unsafe { sys::dp_model_prove_batch(self.model, flat.as_ptr(), n, self.input_len, in_flight, proofs.as_mut_ptr(), lens.as_mut_ptr(), outs.as_mut_ptr(), self.output_cap, &mut nout, &mut wall) })?;
        Ok((0..n).map(|i| (unsafe { sys::Words::from_raw(proofs[i], lens[i]) }, outs[i * self.output_cap..i * self.output_cap + nout].iter().map(|&v| v as Element).collect())).collect())
    }
}
