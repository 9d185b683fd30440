// The bodies the RESIDENT EXECUTOR (rx.hip) can run, in id order — ONE list read by both translation units: hip_dev.hip turns a
// launch of Body into (id, argument pack) with rx_body_id<Body>(), rx.hip's workers switch on the id. Everything dp_model_prove_batch
// launches in throughput mode is here; a body that is missing is refused at submit time ("not available in the resident executor").
// X(class, body): class RX_STREAM = grid-stride / wide bodies (<= 20 KB of LDS, <= 96 VGPRs: several workers per CU),
//                 class RX_BIG    = one-workgroup protocol bodies and the LDS-tiled passes (64 KB of LDS, 128 VGPRs: one worker per CU).
#pragma once
#define RX_STREAM 0
#define RX_BIG 1
#define RX_BODY_LIST(X) \
  X(RX_STREAM, k_copy_words) X(RX_STREAM, k_download) X(RX_STREAM, k_zero_words) X(RX_STREAM, k_fieldize) X(RX_STREAM, k_publish) \
  X(RX_STREAM, k_eq_table) X(RX_STREAM, k_eq_table_many) X(RX_STREAM, k_eq_outer_many) X(RX_STREAM, k_mle_eval_partial) X(RX_STREAM, k_reduce_publish) \
  X(RX_STREAM, k_fix_high_partial) X(RX_STREAM, k_colsum) X(RX_STREAM, k_fix_low) X(RX_STREAM, k_fold) X(RX_STREAM, k_finish_publish) \
  X(RX_STREAM, k_sc_terms<false>) X(RX_BIG, k_sc_terms<true>) \
  X(RX_STREAM, k_sc_fused<1, true, true>) X(RX_STREAM, k_sc_fused<1, true, false>) X(RX_STREAM, k_sc_fused<1, false, true>) X(RX_STREAM, k_sc_fused<1, false, false>) \
  X(RX_STREAM, k_sc_fused<2, true, true>) X(RX_STREAM, k_sc_fused<2, true, false>) X(RX_STREAM, k_sc_fused<2, false, true>) X(RX_STREAM, k_sc_fused<2, false, false>) \
  X(RX_STREAM, k_sc_fused<3, true, true>) X(RX_STREAM, k_sc_fused<3, true, false>) X(RX_STREAM, k_sc_fused<3, false, true>) X(RX_STREAM, k_sc_fused<3, false, false>) \
  X(RX_STREAM, k_logup_den) X(RX_STREAM, k_logup_layer) X(RX_STREAM, k_logup_tree) \
  X(RX_STREAM, k_mobius_stage<false>) X(RX_STREAM, k_mobius_stage<true>) X(RX_STREAM, k_rs_prepare<false>) X(RX_STREAM, k_rs_prepare<true>) \
  X(RX_STREAM, k_ntt_stage<false>) X(RX_STREAM, k_ntt_stage<true>) X(RX_STREAM, k_bitrev<false>) X(RX_STREAM, k_bitrev<true>) \
  X(RX_STREAM, k_merkle_leaves<false>) X(RX_STREAM, k_merkle_leaves<true>) X(RX_STREAM, k_merkle_layer) X(RX_STREAM, k_merkle_layer_lp) X(RX_STREAM, k_merkle_layers) \
  X(RX_STREAM, k_merkle_leaves_many<false>) X(RX_STREAM, k_merkle_leaves_many<true>) X(RX_STREAM, k_merkle_layer_many) \
  X(RX_STREAM, k_ntt_stage_many) X(RX_STREAM, k_bitrev_many) \
  X(RX_STREAM, k_classic_fused) X(RX_STREAM, k_classic_reduce) X(RX_STREAM, k_classic_fold) X(RX_STREAM, k_classic_sums) \
  X(RX_STREAM, k_axpy_many) X(RX_STREAM, k_axpy_rep) X(RX_STREAM, k_bf_msg) X(RX_STREAM, k_fri_fold) X(RX_STREAM, k_query_gather) \
  X(RX_BIG, k_sc_small<false>) X(RX_BIG, k_sc_small<true>) X(RX_BIG, k_sc_persist<false>) X(RX_BIG, k_sc_persist<true>) \
  X(RX_BIG, k_sc_persist_lds<false>) X(RX_BIG, k_sc_persist_lds<true>) \
  X(RX_BIG, k_logup_tail) X(RX_BIG, k_classic_tail) X(RX_BIG, k_dense_tail) X(RX_BIG, k_eqsum_tail) X(RX_BIG, k_deleg_tail) X(RX_BIG, k_commit_tail) X(RX_BIG, k_merkle_tail) \
  X(RX_BIG, k_butterfly_pass<false, false>) X(RX_BIG, k_butterfly_pass<false, true>) X(RX_BIG, k_butterfly_pass<true, false>) X(RX_BIG, k_butterfly_pass<true, true>) \
  X(RX_BIG, k_commit_small<false>) X(RX_BIG, k_commit_small<true>)
