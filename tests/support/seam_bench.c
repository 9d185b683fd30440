/* TEST / BENCH HARNESS (tests/ and bench.py only): what does a host get that proves THROUGH THE SEAMS — the entry points a Rust
 * `Prover::<_, _, BasefoldHip>` with a patched `prove_parallel` would bind (seam 1: dp_pcs_commit / dp_pcs_batch_open; seam 2:
 * dp_sumcheck_prove, dp_logup_prove; the table primitives dp_buf_upload / dp_mle_fix_high / dp_mle_eval) — instead of handing the
 * whole model to dp_model_prove_batch? T host threads, each with its own dp_ctx (blocking calls: modes 0 / 2), or ONE host thread that keeps
 * T proofs in flight through the submit / poll forms (dp_async, mode 3), proving a stream of "proofs" by issuing, call for call and shape for
 * shape, the seam calls of one Dense-4M proof (zkml/src/iop/prover.rs:401-488 over the layers of mlp.py 5 x 1024):
 *   witness        31 x (dp_buf_upload of a 2^10-row column + dp_pcs_commit), 1 table of 2^15, 2 of 2^8      commit/context.rs, lookup/context.rs:631-781
 *   per Dense (6)  dp_mle_fix_high on the committed 2^20 weights, one degree-2 sumcheck over 2^10, dp_mle_eval of the bias   layers/dense.rs:423-561
 *   per Requant(6) two lookup logup-GKR proofs (2 columns of 2^10 each) + a 3-table degree-2 sumcheck over 2^10              layers/requant.rs:531-690
 *   per ReLU (5)   one logup-GKR proof (2 columns) + the same_poly sumcheck                                                  layers/activation.rs:385-456
 *   tables (3)     logup-GKR in table mode (multiplicities)                                                                  iop/prover.rs:110-157
 *   opening        dp_pcs_batch_open over the 4 weight commitments (2^20), the first layer's (2^12), the 2^15 table and the 31 columns
 * The tables hold random field elements (the prover's work does not depend on the values; the proofs are NOT checked here — the
 * seams' bit-exactness is what tests/test_gpu_primitives.py, test_gpu_c_consumer.py and test_gpu_zzzz_rx.py establish). The figure
 * is therefore "seam-level, workload-equivalent proofs per second", to be read next to dp_model_prove_batch's rate.
 * usage: seam_bench <threads> <proofs per thread> [0: plain contexts | 2: plain contexts in throughput mode (dp_ctx_set_throughput_mode) |
 * 3: ONE thread, <threads> proofs in flight on one context through dp_async (submit / poll) |
 * 4: <threads> threads making the BLOCKING calls of mode 0 on ONE shared context routed to one engine (dp_ctx_route_to_engine): what a host written against
 *    the reference's synchronous traits gets — every call is a submit + wait, calls of the same shape from different threads are merged] */
#define _POSIX_C_SOURCE 200809L
#include "../../include/deep_prove_hip.h"
#include <pthread.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#define CHECK(x) do { int32_t rc_ = (x); if (rc_ != DP_OK) { fprintf(stderr, "%s failed: [%d] %s\n", #x, (int)rc_, dp_last_error()); exit(1); } } while (0)
#define P 0xFFFFFFFF00000001ull
enum { NV = 10, N = 1 << NV, NCOLS = 31, NDENSE = 6, NREQ = 6, NRELU = 5 };

static uint64_t splitmix(uint64_t* s) { uint64_t z = (*s += 0x9E3779B97F4A7C15ull); z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull; z = (z ^ (z >> 27)) * 0x94D049BB133111EBull; return z ^ (z >> 31); }
static void fill(uint64_t* w, size_t n, uint64_t* s) { for (size_t i = 0; i < n; i++) w[i] = splitmix(s) % P; }
static double now_s(void) { struct timespec t; clock_gettime(CLOCK_MONOTONIC, &t); return (double)t.tv_sec + 1e-9 * (double)t.tv_nsec; }

struct worker {
  int id, proofs, use_executor;
  dp_ctx* ctx;
  dp_buf* weights[4]; dp_commit* wcomm[4];         /* 1024 x 1024 base matrices (Context::generate: outside the timed region) */
  dp_buf* w0; dp_commit* w0comm;                   /* the 1024 x 4 first layer */
  dp_buf* bias;                                    /* 2^10 */
  dp_buf* big_table; dp_commit* big_comm;          /* 2^15-row clamping table multiplicities */
  uint64_t* col_words; uint64_t* ext_words;        /* host data of a column / an extension table */
  uint64_t point20[40], point12[24], point15[30], point10[20];
  double seconds;
};

static pthread_barrier_t g_start;

static void one_proof(struct worker* w) {
  dp_ctx* ctx = w->ctx;
  dp_transcript* t = dp_transcript_new("m2vec");
  dp_buf* cols[NCOLS]; dp_commit* comms[NCOLS];
  uint64_t root[4];
  /* witness columns: uploaded and committed one by one, as PCS::commit is called per column */
  for (int i = 0; i < NCOLS; i++) { CHECK(dp_buf_upload(ctx, w->col_words, N, 0, &cols[i])); CHECK(dp_pcs_commit(ctx, cols[i], &comms[i], root)); CHECK(dp_transcript_append_elements(t, root, 4)); }
  dp_buf* e0; dp_buf* e1; dp_buf* e2;
  CHECK(dp_buf_upload(ctx, w->ext_words, N, 1, &e0)); CHECK(dp_buf_upload(ctx, w->ext_words + 2 * N, N, 1, &e1)); CHECK(dp_buf_upload(ctx, w->ext_words + 4 * N, N, 1, &e2));
  uint64_t* pw = NULL; size_t pn = 0; uint64_t finals[6], ev[2];
  const uint64_t cc[2] = {12345, 678}, csc[2] = {91011, 1213};
  int col = 0;
  for (int l = 0; l < NDENSE; l++) {
    /* Dense: fix the row variables of the weights at the claim's point, one degree-2 sumcheck, the bias evaluation */
    dp_buf* fixed = NULL;
    CHECK(dp_mle_fix_high(ctx, w->weights[l % 4], N, N, w->point10, &fixed));
    { const dp_buf* tabs[2] = {fixed, e0}; const int32_t deg[1] = {2}, tt[2] = {0, 1}; const uint64_t co[2] = {1, 0};
      CHECK(dp_sumcheck_prove(ctx, NV, tabs, 2, deg, tt, co, 1, t, &pw, &pn, finals)); dp_free(pw); }
    CHECK(dp_mle_eval(ctx, w->bias, w->point10, NV, ev));
    CHECK(dp_buf_free(ctx, fixed));
    /* Requant: two lookups (2 columns per instance) and the batching sumcheck */
    for (int k = 0; k < 2; k++) { const dp_buf* lc[2] = {cols[col % NCOLS], cols[(col + 1) % NCOLS]}; col += 2; CHECK(dp_logup_prove(ctx, lc, 2, 2, NULL, cc, csc, t, &pw, &pn)); dp_free(pw); }
    { const dp_buf* tabs[3] = {e0, e1, e2}; const int32_t deg[2] = {2, 2}, tt[4] = {0, 1, 0, 2}; const uint64_t co[4] = {3, 4, 5, 6};
      CHECK(dp_sumcheck_prove(ctx, NV, tabs, 3, deg, tt, co, 2, t, &pw, &pn, finals)); dp_free(pw); }
    if (l < NRELU) {  /* ReLU: one lookup + the same_poly accumulation sumcheck */
      const dp_buf* lc[2] = {cols[col % NCOLS], cols[(col + 1) % NCOLS]}; col += 2;
      CHECK(dp_logup_prove(ctx, lc, 2, 2, NULL, cc, csc, t, &pw, &pn)); dp_free(pw);
      const dp_buf* tabs[2] = {e1, e2}; const int32_t deg[1] = {2}, tt[2] = {0, 1}; const uint64_t co[2] = {1, 0};
      CHECK(dp_sumcheck_prove(ctx, NV, tabs, 2, deg, tt, co, 1, t, &pw, &pn, finals)); dp_free(pw);
    }
  }
  /* the lookup tables' own logup proofs: a 2^15-row table with its multiplicities, two of 2^8 */
  { const dp_buf* tc[1] = {w->big_table}; CHECK(dp_logup_prove(ctx, tc, 1, 1, w->big_table, cc, csc, t, &pw, &pn)); dp_free(pw); }
  /* the opening: every committed polynomial at one point of its size */
  { enum { NC = 4 + 1 + 1 + NCOLS };
    const dp_commit* cm[NC]; uint64_t pts[4 * 40 + 24 + 30 + NCOLS * 20]; uint64_t evals[2 * NC]; size_t o = 0; int c = 0;
    for (int i = 0; i < 4; i++) { cm[c++] = w->wcomm[i]; memcpy(pts + o, w->point20, 320); o += 40; }
    cm[c++] = w->w0comm; memcpy(pts + o, w->point12, 192); o += 24;
    cm[c++] = w->big_comm; memcpy(pts + o, w->point15, 240); o += 30;
    for (int i = 0; i < NCOLS; i++) { cm[c++] = comms[i]; memcpy(pts + o, w->point10, 160); o += 20; }
    for (int i = 0; i < 2 * NC; i++) evals[i] = (uint64_t)(i + 1);  /* (claimed values: the prover's work does not depend on them) */
    CHECK(dp_pcs_batch_open(ctx, cm, NC, pts, evals, t, &pw, &pn)); dp_free(pw); }
  for (int i = 0; i < NCOLS; i++) { CHECK(dp_pcs_commit_free(ctx, comms[i])); CHECK(dp_buf_free(ctx, cols[i])); }
  CHECK(dp_buf_free(ctx, e0)); CHECK(dp_buf_free(ctx, e1)); CHECK(dp_buf_free(ctx, e2));
  dp_transcript_free(t);
}


/* ---- mode 3: one host thread, `T` proofs in flight through the submit / poll forms. A proof is a list of steps; the blocking table primitives
 * (dp_mle_fix_high, dp_mle_eval: tens of microseconds on the client's context) run inline, every seam call is a ticket. */
enum { ST_COMMITS, ST_FIX, ST_FIXW, ST_SC2, ST_SC2W, ST_EVAL, ST_EVALW, ST_LOGUP, ST_SC3, ST_SC2B, ST_TABLE, ST_OPEN, ST_DONE };
struct aproof {
  dp_transcript* t; dp_buf* cols[NCOLS]; dp_commit* comms[NCOLS]; dp_ticket* ctk[NCOLS]; int ncommitted;
  dp_buf *e0, *e1, *e2, *fixed; dp_ticket* tk; int step, layer, sub, col, left;
};
static double g_t_begin, g_t_end, g_t_sync;  /* client-thread seconds in uploads + commit submits, in frees, in the blocking table primitives */
static void aproof_begin(struct worker* w, struct aproof* p, dp_async* eng) {
  const double tb0 = now_s();
  p->t = dp_transcript_new("m2vec"); p->tk = NULL; p->layer = 0; p->sub = 0; p->col = 0; p->fixed = NULL; p->ncommitted = 0;
  for (int i = 0; i < NCOLS; i++) CHECK(dp_pcs_commit_host_submit(eng, w->col_words, N, 0, &p->ctk[i]));  /* PCS::commit(&poly): upload + commit, one ticket per column */
  CHECK(dp_buf_upload(w->ctx, w->ext_words, N, 1, &p->e0)); CHECK(dp_buf_upload(w->ctx, w->ext_words + 2 * N, N, 1, &p->e1)); CHECK(dp_buf_upload(w->ctx, w->ext_words + 4 * N, N, 1, &p->e2));
  p->step = ST_COMMITS;
  g_t_begin += now_s() - tb0;
}
static void aproof_end(struct worker* w, struct aproof* p) {
  const double te0 = now_s();
  for (int i = 0; i < NCOLS; i++) { CHECK(dp_pcs_commit_free(w->ctx, p->comms[i])); CHECK(dp_buf_free(w->ctx, p->cols[i])); }
  CHECK(dp_buf_free(w->ctx, p->e0)); CHECK(dp_buf_free(w->ctx, p->e1)); CHECK(dp_buf_free(w->ctx, p->e2));
  dp_transcript_free(p->t);
  g_t_end += now_s() - te0;
}
/* advance as far as possible without blocking; returns 1 when the proof has finished */
static int aproof_step(struct worker* w, struct aproof* p, dp_async* eng) {
  static const uint64_t cc[2] = {12345, 678}, csc[2] = {91011, 1213};
  for (;;) {
    if (p->step == ST_COMMITS) {
      while (p->ncommitted < NCOLS) {  /* roots enter the transcript in column order */
        int s = dp_poll(p->ctk[p->ncommitted]);
        if (s == 0) return 0;
        if (s < 0) { fprintf(stderr, "commit ticket failed: [%d] %s\n", s, dp_last_error()); exit(1); }
        uint64_t root[4];
        CHECK(dp_ticket_buf(p->ctk[p->ncommitted], &p->cols[p->ncommitted]));
        CHECK(dp_ticket_commit(p->ctk[p->ncommitted], &p->comms[p->ncommitted], root)); CHECK(dp_ticket_free(p->ctk[p->ncommitted]));
        CHECK(dp_transcript_append_elements(p->t, root, 4));
        p->ncommitted++;
      }
      p->step = ST_FIX;
      continue;
    }
    if (p->tk) {  /* a seam call in flight */
      int s = dp_poll(p->tk);
      if (s == 0) return 0;
      if (s < 0) { fprintf(stderr, "ticket failed at step %d: [%d] %s\n", p->step, s, dp_last_error()); exit(1); }
      if (p->step == ST_FIXW) { CHECK(dp_ticket_buf(p->tk, &p->fixed)); }
      else if (p->step == ST_EVALW) { uint64_t ev[2]; CHECK(dp_ticket_values(p->tk, ev, 2)); }
      else { uint64_t* pw = NULL; size_t pn = 0; CHECK(dp_ticket_words(p->tk, 0, &pw, &pn)); dp_free(pw); }
      CHECK(dp_ticket_free(p->tk)); p->tk = NULL;
      /* what follows the call that has just completed */
      if (p->step == ST_FIXW) p->step = ST_SC2;
      else if (p->step == ST_SC2W) p->step = ST_EVAL;
      else if (p->step == ST_EVALW) { const double ts0 = now_s(); CHECK(dp_buf_free(w->ctx, p->fixed)); p->fixed = NULL; g_t_sync += now_s() - ts0; p->sub = 0; p->step = ST_LOGUP; }
      else if (p->step == ST_LOGUP) { if (++p->sub < 2) p->step = ST_LOGUP; else p->step = ST_SC3; }
      else if (p->step == ST_SC3) { p->sub = 0; p->step = p->layer < NRELU ? ST_SC2B : ST_FIX; if (p->step == ST_FIX) p->layer++; }
      else if (p->step == ST_SC2B) { if (p->sub == 0) { p->sub = 1; } else { p->sub = 0; p->layer++; p->step = ST_FIX; } }
      else if (p->step == ST_TABLE) p->step = ST_OPEN;
      else if (p->step == ST_OPEN) { p->step = ST_DONE; return 1; }
      if (p->step == ST_FIX && p->layer >= NDENSE) p->step = ST_TABLE;
      continue;
    }
    switch (p->step) {
      case ST_FIX:  /* Dense: the weights with their row variables fixed at the claim's point (a ticket) ... */
        CHECK(dp_mle_fix_high_submit(eng, w->weights[p->layer % 4], N, N, w->point10, &p->tk));
        p->step = ST_FIXW;
        break;
      case ST_SC2: {  /* ... the degree-2 sumcheck over it ... */
        const dp_buf* tabs[2] = {p->fixed, p->e0}; const int32_t deg[1] = {2}, tt[2] = {0, 1}; const uint64_t co[2] = {1, 0};
        CHECK(dp_sumcheck_prove_submit(eng, NV, tabs, 2, deg, tt, co, 1, p->t, &p->tk));
        p->step = ST_SC2W;
        break; }
      case ST_EVAL:   /* ... and the bias evaluation */
        CHECK(dp_mle_eval_submit(eng, w->bias, w->point10, NV, &p->tk));
        p->step = ST_EVALW;
        break;
      case ST_LOGUP: {
        const dp_buf* lc[2] = {p->cols[p->col % NCOLS], p->cols[(p->col + 1) % NCOLS]}; p->col += 2;
        CHECK(dp_logup_prove_submit(eng, lc, 2, 2, NULL, cc, csc, p->t, &p->tk));
        break; }
      case ST_SC3: {
        const dp_buf* tabs[3] = {p->e0, p->e1, p->e2}; const int32_t deg[2] = {2, 2}, tt[4] = {0, 1, 0, 2}; const uint64_t co[4] = {3, 4, 5, 6};
        CHECK(dp_sumcheck_prove_submit(eng, NV, tabs, 3, deg, tt, co, 2, p->t, &p->tk));
        break; }
      case ST_SC2B:
        if (p->sub == 0) {  /* ReLU: the lookup ... */
          const dp_buf* lc[2] = {p->cols[p->col % NCOLS], p->cols[(p->col + 1) % NCOLS]}; p->col += 2;
          CHECK(dp_logup_prove_submit(eng, lc, 2, 2, NULL, cc, csc, p->t, &p->tk));
        } else {            /* ... and the same_poly accumulation sumcheck */
          const dp_buf* tabs[2] = {p->e1, p->e2}; const int32_t deg[1] = {2}, tt[2] = {0, 1}; const uint64_t co[2] = {1, 0};
          CHECK(dp_sumcheck_prove_submit(eng, NV, tabs, 2, deg, tt, co, 1, p->t, &p->tk));
        }
        break;
      case ST_TABLE: {
        const dp_buf* tc[1] = {w->big_table};
        CHECK(dp_logup_prove_submit(eng, tc, 1, 1, w->big_table, cc, csc, p->t, &p->tk));
        break; }
      case ST_OPEN: {
        enum { NC = 4 + 1 + 1 + NCOLS };
        const dp_commit* cm[NC]; uint64_t pts[4 * 40 + 24 + 30 + NCOLS * 20]; uint64_t evals[2 * NC]; size_t o = 0; int c = 0;
        for (int i = 0; i < 4; i++) { cm[c++] = w->wcomm[i]; memcpy(pts + o, w->point20, 320); o += 40; }
        cm[c++] = w->w0comm; memcpy(pts + o, w->point12, 192); o += 24;
        cm[c++] = w->big_comm; memcpy(pts + o, w->point15, 240); o += 30;
        for (int i = 0; i < NCOLS; i++) { cm[c++] = p->comms[i]; memcpy(pts + o, w->point10, 160); o += 20; }
        for (int i = 0; i < 2 * NC; i++) evals[i] = (uint64_t)(i + 1);
        CHECK(dp_pcs_batch_open_submit(eng, cm, NC, pts, evals, p->t, &p->tk));
        break; }
      default: return 1;
    }
  }
}
static void worker_setup(struct worker* w);
static int run_async(int T, int per) {
  struct worker w; memset(&w, 0, sizeof w); w.id = 0;
  worker_setup(&w);
  dp_async* eng = NULL;
  CHECK(dp_async_create(w.ctx, 2 * T > 64 ? 2 * T : 64, 0, &eng));
  struct aproof* ps = (struct aproof*)calloc((size_t)T, sizeof *ps);
  /* warm-up: one proof */
  aproof_begin(&w, &ps[0], eng); while (!aproof_step(&w, &ps[0], eng)) {} aproof_end(&w, &ps[0]);
  g_t_begin = g_t_end = g_t_sync = 0;
  const double t0 = now_s();
  int started = 0, finished = 0; const int total = T * per;
  for (int i = 0; i < T && started < total; i++) { aproof_begin(&w, &ps[i], eng); ps[i].left = 1; started++; }
  while (finished < total) {
    for (int i = 0; i < T; i++) {
      if (!ps[i].left) continue;
      if (aproof_step(&w, &ps[i], eng)) {
        aproof_end(&w, &ps[i]); finished++; ps[i].left = 0;
        if (started < total) { aproof_begin(&w, &ps[i], eng); ps[i].left = 1; started++; }
      }
    }
  }
  const double dt = now_s() - t0;
  size_t calls = 0, groups = 0, merged = 0, workers = 0;
  CHECK(dp_async_stats(eng, &calls, &groups, &merged, &workers));
  CHECK(dp_async_destroy(eng));
  printf("{\"seam_level_proofs_per_s\": %.2f, \"threads\": 1, \"proofs_in_flight\": %d, \"proofs\": %d, \"seconds\": %.3f, \"async\": true, \"calls\": %zu, \"groups\": %zu, \"calls_merged\": %zu, \"workers\": %zu, "
         "\"client_thread_s\": {\"uploads_and_commit_submits\": %.3f, \"frees\": %.3f, \"frees_of_fixed_tables\": %.3f}}\n",
         (double)total / dt, T, total, dt, calls, groups, merged, workers, g_t_begin, g_t_end, g_t_sync);
  return 0;
}

static void worker_setup(struct worker* w) {
  uint64_t seed = 0xD33B0000ull + (uint64_t)w->id;
  CHECK(dp_ctx_create(0, &w->ctx));
  CHECK(dp_pcs_setup(w->ctx, (size_t)1 << 20));
  uint64_t* big = (uint64_t*)malloc(8u << 20);
  for (int i = 0; i < 4; i++) { fill(big, (size_t)1 << 20, &seed); CHECK(dp_buf_upload(w->ctx, big, (size_t)1 << 20, 0, &w->weights[i])); uint64_t r[4]; CHECK(dp_pcs_commit(w->ctx, w->weights[i], &w->wcomm[i], r)); }
  { fill(big, (size_t)1 << 12, &seed); CHECK(dp_buf_upload(w->ctx, big, (size_t)1 << 12, 0, &w->w0)); uint64_t r[4]; CHECK(dp_pcs_commit(w->ctx, w->w0, &w->w0comm, r)); }
  { fill(big, (size_t)1 << 15, &seed); CHECK(dp_buf_upload(w->ctx, big, (size_t)1 << 15, 0, &w->big_table)); uint64_t r[4]; CHECK(dp_pcs_commit(w->ctx, w->big_table, &w->big_comm, r)); }
  { fill(big, N, &seed); CHECK(dp_buf_upload(w->ctx, big, N, 0, &w->bias)); }
  free(big);
  w->col_words = (uint64_t*)malloc(8 * N); fill(w->col_words, N, &seed);
  w->ext_words = (uint64_t*)malloc(8 * 6 * N); fill(w->ext_words, 6 * N, &seed);
  fill(w->point20, 40, &seed); fill(w->point12, 24, &seed); fill(w->point15, 30, &seed); fill(w->point10, 20, &seed);
}

/* ---- mode 4: T threads, blocking calls, ONE context whose seam calls are routed to one engine */
static struct worker g_shared;
static void* run_routed(void* arg) {
  struct worker* w = (struct worker*)arg;
  pthread_barrier_wait(&g_start);
  one_proof(w);  /* warm-up */
  pthread_barrier_wait(&g_start);
  const double t0 = now_s();
  for (int i = 0; i < w->proofs; i++) one_proof(w);
  w->seconds = now_s() - t0;
  pthread_barrier_wait(&g_start);
  return NULL;
}
static int run_routed_main(int T, int per) {
  memset(&g_shared, 0, sizeof g_shared);
  worker_setup(&g_shared);
  dp_async* eng = NULL;
  CHECK(dp_async_create(g_shared.ctx, T > 64 ? T : 64, 0, &eng));
  CHECK(dp_ctx_route_to_engine(g_shared.ctx, eng));
  pthread_barrier_init(&g_start, NULL, (unsigned)T + 1);
  struct worker* ws = (struct worker*)calloc((size_t)T, sizeof *ws);
  pthread_t* th = (pthread_t*)calloc((size_t)T, sizeof *th);
  for (int i = 0; i < T; i++) { ws[i] = g_shared; ws[i].id = i; ws[i].proofs = per; pthread_create(&th[i], NULL, run_routed, &ws[i]); }
  pthread_barrier_wait(&g_start);
  pthread_barrier_wait(&g_start);  /* warm-up proofs done */
  const double t0 = now_s();
  pthread_barrier_wait(&g_start);
  const double dt = now_s() - t0;
  for (int i = 0; i < T; i++) pthread_join(th[i], NULL);
  size_t calls = 0, groups = 0, merged = 0, workers = 0;
  CHECK(dp_async_stats(eng, &calls, &groups, &merged, &workers));
  CHECK(dp_ctx_route_to_engine(g_shared.ctx, NULL));
  CHECK(dp_async_destroy(eng));
  printf("{\"seam_level_proofs_per_s\": %.2f, \"threads\": %d, \"proofs\": %d, \"seconds\": %.3f, \"blocking_calls_routed_to_engine\": true, \"calls\": %zu, \"groups\": %zu, \"calls_merged\": %zu, \"workers\": %zu, "
         "\"ms_per_proof_per_thread\": %.1f}\n", (double)T * per / dt, T, T * per, dt, calls, groups, merged, workers, 1000.0 * dt / per);
  return 0;
}

static void* run(void* arg) {
  struct worker* w = (struct worker*)arg;
  worker_setup(w);
  pthread_barrier_wait(&g_start);  /* every context exists (the library's code objects are loaded, PCS::setup has run) ... */
  pthread_barrier_wait(&g_start);
  if (w->use_executor == 2) CHECK(dp_ctx_set_throughput_mode(w->ctx, 1));  /* plain contexts, fused device-side Fiat-Shamir kernels */
  one_proof(w);  /* warm-up */
  pthread_barrier_wait(&g_start);
  const double t0 = now_s();
  for (int i = 0; i < w->proofs; i++) one_proof(w);
  w->seconds = now_s() - t0;
  pthread_barrier_wait(&g_start);
  return NULL;
}

int main(int argc, char** argv) {
  const int T = argc > 1 ? atoi(argv[1]) : 8, per = argc > 2 ? atoi(argv[2]) : 4, use_executor = argc > 3 ? atoi(argv[3]) : 0;
  if (T < 1 || T > 1024 || per < 1) { fprintf(stderr, "usage: seam_bench <threads | proofs in flight> <proofs per thread> [0 | 2 | 3 | 4]\n"); return 2; }
  if (use_executor == 3) return run_async(T, per);
  if (use_executor == 4) return run_routed_main(T, per);
  pthread_barrier_init(&g_start, NULL, (unsigned)T + 1);
  struct worker* ws = (struct worker*)calloc((size_t)T, sizeof *ws);
  pthread_t* th = (pthread_t*)calloc((size_t)T, sizeof *th);
  for (int i = 0; i < T; i++) { ws[i].id = i; ws[i].proofs = per; ws[i].use_executor = use_executor; pthread_create(&th[i], NULL, run, &ws[i]); }
  pthread_barrier_wait(&g_start);  /* contexts ready */
  pthread_barrier_wait(&g_start);
  pthread_barrier_wait(&g_start);  /* warm-up proofs done */
  const double t0 = now_s();
  pthread_barrier_wait(&g_start);
  const double dt = now_s() - t0;
  for (int i = 0; i < T; i++) pthread_join(th[i], NULL);
  printf("{\"seam_level_proofs_per_s\": %.2f, \"threads\": %d, \"proofs\": %d, \"seconds\": %.3f, \"throughput_mode\": %s, \"ms_per_proof_per_thread\": %.1f}\n",
         (double)T * per / dt, T, T * per, dt, use_executor == 2 ? "true" : "false", 1000.0 * dt / per);
  return 0;
}
