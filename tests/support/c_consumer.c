/* TEST HARNESS (tests/ only): a C11 consumer of include/deep_prove_hip.h — what bindgen / cgo / JNI would see. It includes
 * the header as C, links libdeepprove_hip.so and drives the drop-in seams from plain C with pthreads:
 *   seam 2  dp_sumcheck_prove (a degree-4 product + a short table) -> sc_proof.bin, sc_finals.bin
 *   seam 1  dp_pcs_setup, dp_pcs_commit from FOUR THREADS AT ONCE on one dp_ctx (as rayon calls PCS::commit) -> roots.bin,
 *           dp_pcs_commitment, dp_mle_eval, dp_pcs_batch_open -> bo_proof.bin, dp_pcs_batch_verify,
 *           dp_pcs_open / dp_pcs_verify on a 6-variable polynomial -> triv_proof.bin, on the 13-variable one -> open_proof.bin
 *   model   dp_model_setup / dp_model_output_len / dp_model_prove / dp_model_verifier_blob / dp_verify -> model_proof.bin
 * Inputs are files written by tests/test_gpu_c_consumer.py, which compares every output with the oracle.
 * usage: c_consumer <workdir>        (c_consumer --symbols: only touch every entry point's address, no device needed) */
#include "../../include/deep_prove_hip.h"
#include <pthread.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#define CHECK(x) do { int32_t rc_ = (x); if (rc_ != DP_OK) { fprintf(stderr, "%s failed: [%d] %s\n", #x, (int)rc_, dp_last_error()); exit(1); } } while (0)

static char g_dir[1024];
static uint64_t* read_u64(const char* name, size_t* n) {
  char path[1200]; snprintf(path, sizeof path, "%s/%s", g_dir, name);
  FILE* f = fopen(path, "rb"); if (!f) { fprintf(stderr, "cannot open %s\n", path); exit(1); }
  fseek(f, 0, SEEK_END); long sz = ftell(f); fseek(f, 0, SEEK_SET);
  uint64_t* p = (uint64_t*)malloc((size_t)sz ? (size_t)sz : 8);
  if (fread(p, 1, (size_t)sz, f) != (size_t)sz) { fprintf(stderr, "short read %s\n", path); exit(1); }
  fclose(f); *n = (size_t)sz / 8; return p;
}
static void write_u64(const char* name, const uint64_t* p, size_t n) {
  char path[1200]; snprintf(path, sizeof path, "%s/%s", g_dir, name);
  FILE* f = fopen(path, "wb"); if (!f || fwrite(p, 8, n, f) != n) { fprintf(stderr, "cannot write %s\n", path); exit(1); }
  fclose(f);
}

struct commit_job { dp_ctx* ctx; const uint64_t* words; size_t n; dp_buf* buf; dp_commit* comm; uint64_t root[4]; };
static void* commit_thread(void* arg) {
  struct commit_job* j = (struct commit_job*)arg;
  CHECK(dp_buf_upload(j->ctx, j->words, j->n, 0, &j->buf));
  CHECK(dp_pcs_commit(j->ctx, j->buf, &j->comm, j->root));
  return NULL;
}

typedef void (*fn_t)(void);
static int symbols_only(void) {
  /* every entry point a binding generator would emit, referenced so that the linker must resolve it */
  const fn_t syms[] = {(fn_t)dp_last_error, (fn_t)dp_free, (fn_t)dp_ctx_create, (fn_t)dp_ctx_destroy, (fn_t)dp_ctx_name,
    (fn_t)dp_profile_enable, (fn_t)dp_profile_report, (fn_t)dp_probe_compress_rate, (fn_t)dp_buf_from_i64, (fn_t)dp_buf_upload,
    (fn_t)dp_buf_download, (fn_t)dp_buf_len, (fn_t)dp_buf_is_ext, (fn_t)dp_buf_free, (fn_t)dp_transcript_new, (fn_t)dp_transcript_free,
    (fn_t)dp_transcript_append_elements, (fn_t)dp_transcript_append_message, (fn_t)dp_transcript_challenge, (fn_t)dp_eq_table, (fn_t)dp_mle_eval,
    (fn_t)dp_mle_fix_high, (fn_t)dp_sumcheck_prove, (fn_t)dp_sumcheck_verify, (fn_t)dp_sc_session_new, (fn_t)dp_sc_session_round,
    (fn_t)dp_sc_session_finish, (fn_t)dp_sc_session_free, (fn_t)dp_logup_prove, (fn_t)dp_logup_verify, (fn_t)dp_pcs_setup, (fn_t)dp_pcs_commit,
    (fn_t)dp_pcs_commit_free, (fn_t)dp_pcs_commitment, (fn_t)dp_pcs_open, (fn_t)dp_pcs_verify, (fn_t)dp_pcs_batch_open, (fn_t)dp_pcs_batch_verify,
    (fn_t)dp_pcs_batch_open_evals, (fn_t)dp_pcs_batch_verify_evals, (fn_t)dp_pcs_batch_commit, (fn_t)dp_pcs_batch_commit_free, (fn_t)dp_pcs_simple_batch_open, (fn_t)dp_pcs_simple_batch_verify,
    (fn_t)dp_model_setup, (fn_t)dp_model_free, (fn_t)dp_model_prove, (fn_t)dp_model_prove_batch, (fn_t)dp_model_in_flight, (fn_t)dp_model_output_len,
    (fn_t)dp_host_cpu_budget, (fn_t)dp_host_poseidon2, (fn_t)dp_model_infer_host, (fn_t)dp_model_verifier_blob, (fn_t)dp_verify, (fn_t)dp_verify_batch, (fn_t)dp_dist_unique_id, (fn_t)dp_dist_init, (fn_t)dp_dist_free,
    (fn_t)dp_sumcheck_prove_sharded, (fn_t)dp_sumcheck_prove_sharded_local,
    (fn_t)dp_ctx_set_throughput_mode, (fn_t)dp_async_create, (fn_t)dp_ctx_route_to_engine, (fn_t)dp_async_destroy, (fn_t)dp_async_stats, (fn_t)dp_pcs_commit_submit, (fn_t)dp_sumcheck_prove_submit,
    (fn_t)dp_logup_prove_submit, (fn_t)dp_pcs_batch_open_submit, (fn_t)dp_poll, (fn_t)dp_wait, (fn_t)dp_ticket_words, (fn_t)dp_ticket_values, (fn_t)dp_ticket_commit, (fn_t)dp_ticket_free, (fn_t)dp_mle_fix_high_submit, (fn_t)dp_mle_eval_submit, (fn_t)dp_ticket_buf, (fn_t)dp_pcs_commit_host_submit};
  size_t n = sizeof syms / sizeof syms[0], ok = 0;
  for (size_t i = 0; i < n; i++) ok += syms[i] != NULL;
  printf("c11 consumer: %zu of %zu entry points resolved\n", ok, n);
  return ok == n ? 0 : 1;
}

int main(int argc, char** argv) {
  if (argc > 1 && strcmp(argv[1], "--symbols") == 0) return symbols_only();
  if (argc < 2) { fprintf(stderr, "usage: c_consumer <workdir> | --symbols\n"); return 2; }
  snprintf(g_dir, sizeof g_dir, "%s", argv[1]);
  dp_ctx* ctx = NULL;
  CHECK(dp_ctx_create(0, &ctx));
  printf("device: %s\n", dp_ctx_name(ctx));

  /* ---- seam 2: sum of  c0 * t0*t1*t2*t3  +  c1 * s  over 10 variables, s a 7-variable table */
  {
    size_t n; enum { NV = 10 };
    dp_buf* tabs[5];
    for (int i = 0; i < 4; i++) { char nm[32]; snprintf(nm, sizeof nm, "sc_tab%d.bin", i); uint64_t* w = read_u64(nm, &n); CHECK(dp_buf_upload(ctx, w, (size_t)1 << NV, i == 2, &tabs[i])); free(w); }
    { uint64_t* w = read_u64("sc_short.bin", &n); CHECK(dp_buf_upload(ctx, w, (size_t)1 << 7, 0, &tabs[4])); free(w); }
    const int32_t degree[2] = {4, 1}, term_tables[5] = {0, 1, 2, 3, 4};
    const uint64_t coeffs[4] = {1, 0, 12345, 678};
    dp_transcript* t = dp_transcript_new("test");
    uint64_t* proof = NULL; size_t pn = 0; uint64_t finals[10];
    CHECK(dp_sumcheck_prove(ctx, NV, (const dp_buf* const*)tabs, 5, degree, term_tables, coeffs, 2, t, &proof, &pn, finals));
    write_u64("sc_proof.bin", proof, pn); write_u64("sc_finals.bin", finals, 10);
    uint64_t ch[2]; CHECK(dp_transcript_challenge(t, NULL, ch)); write_u64("sc_after.bin", ch, 2);
    dp_free(proof); dp_transcript_free(t);
    for (int i = 0; i < 5; i++) CHECK(dp_buf_free(ctx, tabs[i]));
  }

  /* ---- seam 1: four commits from four threads on ONE context, then batch_open / batch_verify of two of them */
  {
    CHECK(dp_pcs_setup(ctx, (size_t)1 << 13));
    struct commit_job jobs[4]; pthread_t th[4]; size_t n;
    const char* names[4] = {"poly0.bin", "poly1.bin", "poly2.bin", "poly3.bin"};  /* 2^13, 2^12, 2^10, 2^6 base entries */
    uint64_t* words[4];
    for (int i = 0; i < 4; i++) { words[i] = read_u64(names[i], &n); jobs[i].ctx = ctx; jobs[i].words = words[i]; jobs[i].n = n; jobs[i].buf = NULL; jobs[i].comm = NULL; }
    for (int i = 0; i < 4; i++) pthread_create(&th[i], NULL, commit_thread, &jobs[i]);
    for (int i = 0; i < 4; i++) pthread_join(th[i], NULL);
    uint64_t roots[16];
    for (int i = 0; i < 4; i++) {
      uint64_t r[4]; uint32_t nv; int32_t isb;
      CHECK(dp_pcs_commitment(jobs[i].comm, r, &nv, &isb));
      if (memcmp(r, jobs[i].root, 32) != 0 || !isb || ((size_t)1 << nv) != jobs[i].n) { fprintf(stderr, "dp_pcs_commitment disagrees with dp_pcs_commit\n"); return 1; }
      memcpy(roots + 4 * i, r, 32);
    }
    write_u64("roots.bin", roots, 16);
    /* open poly0 (13 vars) and poly2 (10 vars) at the points of points.bin: evaluations by dp_mle_eval */
    uint64_t* pts = read_u64("points.bin", &n);  /* 13 + 10 extension coordinates */
    uint64_t evals[4];
    CHECK(dp_mle_eval(ctx, jobs[0].buf, pts, 13, evals)); CHECK(dp_mle_eval(ctx, jobs[2].buf, pts + 26, 10, evals + 2));
    write_u64("evals.bin", evals, 4);
    const dp_commit* cm[2] = {jobs[0].comm, jobs[2].comm};
    dp_transcript* t = dp_transcript_new("test");
    uint64_t* proof = NULL; size_t pn = 0;
    CHECK(dp_pcs_batch_open(ctx, cm, 2, pts, evals, t, &proof, &pn));
    write_u64("bo_proof.bin", proof, pn);
    uint64_t vroots[8]; memcpy(vroots, roots, 32); memcpy(vroots + 4, roots + 8, 32);
    const uint32_t vnv[2] = {13, 10}; const int32_t visb[2] = {1, 1};
    dp_transcript* vt = dp_transcript_new("test");
    CHECK(dp_pcs_batch_verify((size_t)1 << 13, vroots, vnv, visb, 2, pts, evals, proof, pn, vt));
    proof[pn / 2] ^= 1;  /* a tampered proof must be rejected with DP_ERR_VERIFY (or refused as malformed), never accepted */
    dp_transcript* vt2 = dp_transcript_new("test");
    int32_t rc = dp_pcs_batch_verify((size_t)1 << 13, vroots, vnv, visb, 2, pts, evals, proof, pn, vt2);
    if (rc == DP_OK) { fprintf(stderr, "tampered batch opening accepted\n"); return 1; }
    dp_free(proof); dp_transcript_free(t); dp_transcript_free(vt); dp_transcript_free(vt2);
    /* the 6-variable polynomial: PCS::open / PCS::verify (trivial opening) */
    uint64_t ev6[2]; CHECK(dp_mle_eval(ctx, jobs[3].buf, pts, 6, ev6));
    uint64_t* tp = NULL; size_t tn = 0;
    CHECK(dp_pcs_open(ctx, jobs[3].comm, pts, 6, ev6, NULL, &tp, &tn));
    write_u64("triv_proof.bin", tp, tn);
    CHECK(dp_pcs_verify((size_t)1 << 13, roots + 12, 6, 1, pts, ev6, tp, tn, NULL));
    ev6[0] ^= 1;
    if (dp_pcs_verify((size_t)1 << 13, roots + 12, 6, 1, pts, ev6, tp, tn, NULL) != DP_ERR_VERIFY) { fprintf(stderr, "wrong evaluation accepted by dp_pcs_verify\n"); return 1; }
    dp_free(tp);
    /* the 13-variable polynomial alone: PCS::open (commit phase + queries on the device) / PCS::verify */
    dp_transcript* ot = dp_transcript_new("test"); dp_transcript* ovt = dp_transcript_new("test");
    uint64_t* op = NULL; size_t on = 0;
    CHECK(dp_pcs_open(ctx, jobs[0].comm, pts, 13, evals, ot, &op, &on));
    write_u64("open_proof.bin", op, on);
    CHECK(dp_pcs_verify((size_t)1 << 13, roots, 13, 1, pts, evals, op, on, ovt));
    evals[0] ^= 1;
    dp_transcript* ovt2 = dp_transcript_new("test");
    if (dp_pcs_verify((size_t)1 << 13, roots, 13, 1, pts, evals, op, on, ovt2) != DP_ERR_VERIFY) { fprintf(stderr, "wrong evaluation accepted by dp_pcs_verify (13 variables)\n"); return 1; }
    evals[0] ^= 1;
    if (dp_pcs_open(ctx, jobs[0].comm, pts, 13, evals, NULL, &tp, &tn) != DP_ERR_ARG) { fprintf(stderr, "dp_pcs_open of a non-trivial commitment needs a transcript\n"); return 1; }
    dp_free(op); dp_transcript_free(ot); dp_transcript_free(ovt); dp_transcript_free(ovt2);
    for (int i = 0; i < 4; i++) { CHECK(dp_pcs_commit_free(ctx, jobs[i].comm)); CHECK(dp_buf_free(ctx, jobs[i].buf)); free(words[i]); }
    free(pts);
  }

  /* ---- the model path */
  {
    size_t nb, ni;
    int64_t* blob = (int64_t*)read_u64("model.bin", &nb);
    int64_t* input = (int64_t*)read_u64("input.bin", &ni);
    dp_model* m = NULL;
    CHECK(dp_model_setup(ctx, blob, nb, &m));
    size_t nout = 0; CHECK(dp_model_output_len(m, &nout));
    int64_t* out = (int64_t*)malloc(nout * 8);
    uint64_t* proof = NULL; size_t pn = 0, no = nout; double ms = 0;
    CHECK(dp_model_prove(m, input, ni, &proof, &pn, out, &no, &ms));
    write_u64("model_proof.bin", proof, pn); write_u64("model_out.bin", (const uint64_t*)out, no);
    uint64_t* vb = NULL; size_t vn = 0;
    CHECK(dp_model_verifier_blob(m, &vb, &vn));
    CHECK(dp_verify(vb, vn, proof, pn, input, ni, out, no));
    out[0] ^= 1;
    if (dp_verify(vb, vn, proof, pn, input, ni, out, no) != DP_ERR_VERIFY) { fprintf(stderr, "wrong output accepted by dp_verify\n"); return 1; }
    printf("model proof: %zu words in %.2f ms\n", pn, ms);
    dp_free(proof); dp_free(vb); free(out); free(blob); free(input);
    CHECK(dp_model_free(m));
  }
  CHECK(dp_ctx_destroy(ctx));
  printf("c11 consumer: ok\n");
  return 0;
}
