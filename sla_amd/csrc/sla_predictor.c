/*
 * sla_predictor.c -- the reference's encode-side predictor / coder API (include/SLAPredictor.h, include/SLACoder.h)
 * over the batched MI355X kernels: one call = one block = a batch of one.  Every call stages its operands in
 * device memory, launches the kernel the whole-file pipeline uses for that stage, and copies the result back; the
 * scalar tails that need glibc's log or x87 long double (code length, Toeplitz solve, shortest path) are the same
 * host functions the pipeline uses (sla_plan.c, sla_ltm.c).  No stage is computed on the host instead of the
 * device, and there is no fallback: without a HIP device every Create returns NULL.
 */
#define _GNU_SOURCE
#include "sla_internal.h"
#include "SLAPredictor.h"
#include "SLACoder.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

#include <hip/hip_runtime_api.h>

#define PREDICTOR_MAX_SAMPLES 16384u        /* analysis window of the LPC kernels (LDS) */

typedef struct { void* ptr; size_t cap; } dbuf_t;

static int dev_ok(void)
{
  int n = 0;
  return hipGetDeviceCount(&n) == hipSuccess && n > 0;
}

static int dres(dbuf_t* b, size_t bytes)
{
  if (bytes == 0) { bytes = 16; }
  if (b->cap >= bytes) { return 0; }
  if (b->ptr != NULL) { (void)hipFree(b->ptr); b->ptr = NULL; b->cap = 0; }
  bytes += bytes / 4 + 256;
  if (hipMalloc(&b->ptr, bytes) != hipSuccess) { b->ptr = NULL; return -1; }
  b->cap = bytes;
  return 0;
}

static void dfree(dbuf_t* b) { if (b->ptr != NULL) { (void)hipFree(b->ptr); } b->ptr = NULL; b->cap = 0; }

static int up(dbuf_t* b, const void* src, size_t bytes)
{
  if (dres(b, bytes) != 0) { return -1; }
  return (bytes == 0 || hipMemcpy(b->ptr, src, bytes, hipMemcpyHostToDevice) == hipSuccess) ? 0 : -1;
}

static int down(void* dst, const void* d_src, size_t bytes)
{
  if (hipDeviceSynchronize() != hipSuccess) { return -1; }
  return (bytes == 0 || hipMemcpy(dst, d_src, bytes, hipMemcpyDeviceToHost) == hipSuccess) ? 0 : -1;
}

/* ------------------------------------------------------------------ LPC: autocorrelation + Levinson-Durbin */

struct SLALPCCalculator {
  uint32_t max_order;
  dbuf_t d_x, d_groups, d_cands, d_out;
};

struct SLALPCCalculator* SLALPCCalculator_Create(uint32_t max_order)
{
  struct SLALPCCalculator* l;
  if (!dev_ok()) { return NULL; }
  l = (struct SLALPCCalculator*)calloc(1, sizeof(*l));
  if (l != NULL) { l->max_order = max_order; }
  return l;
}

void SLALPCCalculator_Destroy(struct SLALPCCalculator* l)
{
  if (l == NULL) { return; }
  dfree(&l->d_x); dfree(&l->d_groups); dfree(&l->d_cands); dfree(&l->d_out);
  free(l);
}

/* {r0, parcor[0..order]} of the sub-ranges `cands` of one window of doubles, through k_lpc */
static int lpc_candidates(struct SLALPCCalculator* l, const double* x, uint32_t n, uint32_t order,
                          const sla_hip_lpc_cand* cands, uint32_t ncands, double* out /* ncands * (order+2) */)
{
  /* candidates per group so that window + r[] fit the LDS budget (as the whole-file pipeline does) */
  const uint32_t O1 = order + 1, O2 = order + 2;
  const size_t room = (SLA_HIP_LDS_BUDGET / 8 > (size_t)n + 4 * 4 * order) ? (SLA_HIP_LDS_BUDGET / 8 - n - 4 * 4 * order) : 0;
  uint32_t cpg = (uint32_t)(room / O1), ngroups, g;
  sla_hip_lpc_group* groups;
  int rc = -1;
  if (cpg == 0) { return -1; }
  if (cpg > ncands) { cpg = ncands; }
  ngroups = (ncands + cpg - 1) / cpg;
  groups = (sla_hip_lpc_group*)calloc(ngroups, sizeof(*groups));
  if (groups == NULL) { return -1; }
  for (g = 0; g < ngroups; g++) {
    groups[g].pcm_off = 0; groups[g].num_samples = n; groups[g].channel = 0; groups[g].win_off = SLA_HIP_NO_WINDOW;
    groups[g].int_shift = 0; groups[g].cand_first = g * cpg;
    groups[g].cand_count = (ncands - g * cpg < cpg) ? (ncands - g * cpg) : cpg;
    groups[g].slot_first = g * cpg;
  }
  if (up(&l->d_x, x, sizeof(double) * n) == 0 && up(&l->d_groups, groups, sizeof(*groups) * ngroups) == 0
      && up(&l->d_cands, cands, sizeof(*cands) * ncands) == 0 && dres(&l->d_out, sizeof(double) * (size_t)ncands * O2) == 0
      && sla_hip_launch_lpc_f64((const double*)l->d_x.ptr, order, (const sla_hip_lpc_group*)l->d_groups.ptr, ngroups, n, cpg,
                                (const sla_hip_lpc_cand*)l->d_cands.ptr, (double*)l->d_out.ptr, NULL) == 0
      && down(out, l->d_out.ptr, sizeof(double) * (size_t)ncands * O2) == 0) {
    rc = 0;
  }
  free(groups);
  return rc;
}

SLAPredictorApiResult SLALPCCalculator_CalculatePARCORCoefDouble(
    struct SLALPCCalculator* l, const double* data, uint32_t num_samples, double* parcor_coef, uint32_t order)
{
  sla_hip_lpc_cand cd;
  double* out;
  if (l == NULL || data == NULL || parcor_coef == NULL) { return SLAPREDICTOR_APIRESULT_INVALID_ARGUMENT; }
  if (order > l->max_order) { return SLAPREDICTOR_APIRESULT_EXCEED_MAX_ORDER; }
  if (order == 0) { parcor_coef[0] = 0.0; return SLAPREDICTOR_APIRESULT_OK; }
  if (num_samples == 0 || num_samples > PREDICTOR_MAX_SAMPLES || order > 255) { return SLAPREDICTOR_APIRESULT_NG; }
  out = (double*)malloc(sizeof(double) * (order + 2));
  if (out == NULL) { return SLAPREDICTOR_APIRESULT_NG; }
  cd.start = 0; cd.len = num_samples;
  if (lpc_candidates(l, data, num_samples, order, &cd, 1, out) != 0) { free(out); return SLAPREDICTOR_APIRESULT_FAILED_TO_CALCULATION; }
  memcpy(parcor_coef, out + 1, sizeof(double) * (order + 1));
  free(out);
  return SLAPREDICTOR_APIRESULT_OK;
}

SLAPredictorApiResult SLALPCCalculator_EstimateCodeLength(
    const double* data, uint32_t num_samples, uint32_t bits_per_sample,
    const double* parcor_coef, uint32_t order, double* length_per_sample)
{
  struct SLALPCCalculator tmp;
  sla_hip_lpc_cand cd;
  double out[3];
  int rc;
  if (data == NULL || parcor_coef == NULL || length_per_sample == NULL) { return SLAPREDICTOR_APIRESULT_INVALID_ARGUMENT; }
  if (!dev_ok() || num_samples == 0 || num_samples > PREDICTOR_MAX_SAMPLES) { return SLAPREDICTOR_APIRESULT_NG; }
  /* the block's energy is the r[0] of the LPC kernel (same serial sum as src/SLAPredictor.c:432-436) */
  memset(&tmp, 0, sizeof(tmp));
  cd.start = 0; cd.len = num_samples;
  rc = lpc_candidates(&tmp, data, num_samples, 1, &cd, 1, out);
  dfree(&tmp.d_x); dfree(&tmp.d_groups); dfree(&tmp.d_cands); dfree(&tmp.d_out);
  if (rc != 0) { return SLAPREDICTOR_APIRESULT_FAILED_TO_CALCULATION; }
  *length_per_sample = slai_code_length(out[0], num_samples, bits_per_sample, parcor_coef, order);
  return SLAPREDICTOR_APIRESULT_OK;
}

/* ------------------------------------------------------------------ PARCOR lattice */

struct SLALPCSynthesizer {
  uint32_t max_order;
  int used;                    /* a block has gone through since the last reset */
  dbuf_t d_in, d_k, d_chunks, d_out;
};

struct SLALPCSynthesizer* SLALPCSynthesizer_Create(uint32_t max_order)
{
  struct SLALPCSynthesizer* s;
  if (!dev_ok()) { return NULL; }
  s = (struct SLALPCSynthesizer*)calloc(1, sizeof(*s));
  if (s != NULL) { s->max_order = max_order; }
  return s;
}

void SLALPCSynthesizer_Destroy(struct SLALPCSynthesizer* s)
{
  if (s == NULL) { return; }
  dfree(&s->d_in); dfree(&s->d_k); dfree(&s->d_chunks); dfree(&s->d_out);
  free(s);
}

SLAPredictorApiResult SLALPCSynthesizer_Reset(struct SLALPCSynthesizer* s)
{
  if (s == NULL) { return SLAPREDICTOR_APIRESULT_INVALID_ARGUMENT; }
  s->used = 0;
  return SLAPREDICTOR_APIRESULT_OK;
}

SLAPredictorApiResult SLALPCSynthesizer_PredictByParcorCoefInt32(
    struct SLALPCSynthesizer* s, const int32_t* data, uint32_t num_samples,
    const int32_t* parcor_coef, uint32_t order, int32_t* residual)
{
  extern uint32_t sla_hip_lattice_chunk_samples(uint32_t order);
  sla_hip_lattice_chunk* ck;
  uint32_t per, nck, i;
  SLAPredictorApiResult res = SLAPREDICTOR_APIRESULT_NG;
  if (s == NULL || data == NULL || parcor_coef == NULL || residual == NULL) { return SLAPREDICTOR_APIRESULT_INVALID_ARGUMENT; }
  if (order > s->max_order) { return SLAPREDICTOR_APIRESULT_EXCEED_MAX_ORDER; }
  if (s->used) { return SLAPREDICTOR_APIRESULT_NG; }             /* continuing a block is not supported: Reset first */
  if (num_samples == 0) { return SLAPREDICTOR_APIRESULT_OK; }
  s->used = 1;
  if (order == 0) { memmove(residual, data, sizeof(int32_t) * num_samples); return SLAPREDICTOR_APIRESULT_OK; }
  if (order > 255) { return SLAPREDICTOR_APIRESULT_NG; }
  per = sla_hip_lattice_chunk_samples(order);
  nck = (num_samples + per - 1) / per;
  ck = (sla_hip_lattice_chunk*)calloc(nck, sizeof(*ck));
  if (ck == NULL) { return SLAPREDICTOR_APIRESULT_NG; }
  for (i = 0; i < nck; i++) {
    ck[i].blk_off = 0; ck[i].blk_len = num_samples; ck[i].chunk_start = i * per;
    ck[i].count = (num_samples - i * per < per) ? (num_samples - i * per) : per;
    ck[i].channel = 0; ck[i].slot = 0; ck[i].int_shift = 0;
  }
  if (up(&s->d_in, data, sizeof(int32_t) * num_samples) == 0 && up(&s->d_k, parcor_coef, sizeof(int32_t) * (order + 1)) == 0
      && up(&s->d_chunks, ck, sizeof(*ck) * nck) == 0 && dres(&s->d_out, sizeof(int32_t) * num_samples) == 0
      && sla_hip_launch_lattice_raw((const int32_t*)s->d_in.ptr, num_samples, order, (const sla_hip_lattice_chunk*)s->d_chunks.ptr, nck,
                                    (const int32_t*)s->d_k.ptr, (int32_t*)s->d_out.ptr, NULL) == 0
      && down(residual, s->d_out.ptr, sizeof(int32_t) * num_samples) == 0) {
    res = SLAPREDICTOR_APIRESULT_OK;
  }
  free(ck);
  return res;
}

/* ------------------------------------------------------------------ long-term analysis */

struct SLALongTermCalculator {
  uint32_t fft_size, max_taps;
  dbuf_t d_in, d_job, d_tw, d_scratch, d_rec;
};

struct SLALongTermCalculator* SLALongTermCalculator_Create(
    uint32_t fft_size, uint32_t max_pitch_period, uint32_t max_num_pitch_candidates, uint32_t max_num_taps)
{
  struct SLALongTermCalculator* c;
  slai_fft_plan* plan;
  double* tw;
  (void)max_num_pitch_candidates;
  /* the device pitch scan is built for the encoder's constants (src/SLAInternal.h: period 256, <= 5 taps) */
  if (!dev_ok() || max_pitch_period != SLAI_LTM_MAX_PERIOD || max_num_taps > SLAI_MAX_TAPS
      || fft_size < 1024 || fft_size > 65536 || (fft_size & (fft_size - 1))) { return NULL; }
  c = (struct SLALongTermCalculator*)calloc(1, sizeof(*c));
  if (c == NULL) { return NULL; }
  c->fft_size = fft_size; c->max_taps = max_num_taps;
  plan = slai_fft_plan_create(fft_size);
  tw = (double*)malloc(sizeof(double) * SLA_HIP_TWIDDLE_DOUBLES(fft_size));
  if (plan == NULL || tw == NULL) { slai_fft_plan_destroy(plan); free(tw); free(c); return NULL; }
  slai_fft_plan_export(plan, tw);
  if (up(&c->d_tw, tw, sizeof(double) * SLA_HIP_TWIDDLE_DOUBLES(fft_size)) != 0) { dfree(&c->d_tw); free(c); c = NULL; }
  slai_fft_plan_destroy(plan); free(tw);
  return c;
}

void SLALongTermCalculator_Destroy(struct SLALongTermCalculator* c)
{
  if (c == NULL) { return; }
  dfree(&c->d_in); dfree(&c->d_job); dfree(&c->d_tw); dfree(&c->d_scratch); dfree(&c->d_rec);
  free(c);
}

SLAPredictorApiResult SLALongTermCalculator_CalculateCoef(
    struct SLALongTermCalculator* c, const int32_t* data, uint32_t num_samples,
    uint32_t* pitch_num_samples, double* ltm_coef, uint32_t num_taps)
{
  sla_hip_acf_job job;
  double rec[SLA_HIP_ACF_RECORD];
  uint32_t slots = 0;
  int rc;
  if (c == NULL || data == NULL || pitch_num_samples == NULL || ltm_coef == NULL) { return SLAPREDICTOR_APIRESULT_INVALID_ARGUMENT; }
  if (!(num_taps & 1u)) { return SLAPREDICTOR_APIRESULT_INVALID_ARGUMENT; }
  if (num_taps > c->max_taps) { return SLAPREDICTOR_APIRESULT_EXCEED_MAX_ORDER; }
  if (2 * (uint64_t)num_samples > c->fft_size) { return SLAPREDICTOR_APIRESULT_INVALID_ARGUMENT; }
  if (num_samples == 0) { return SLAPREDICTOR_APIRESULT_NG; }
  job.blk_off = 0; job.blk_len = num_samples; job.channel = 0;
  if (sizeof(double) * (size_t)c->fft_size > SLA_HIP_LDS_BUDGET) {
    slots = 1;
    if (dres(&c->d_scratch, sizeof(double) * (size_t)c->fft_size) != 0) { return SLAPREDICTOR_APIRESULT_NG; }
  }
  if (up(&c->d_in, data, sizeof(int32_t) * num_samples) != 0 || up(&c->d_job, &job, sizeof(job)) != 0
      || dres(&c->d_rec, sizeof(rec)) != 0
      || sla_hip_launch_ltm_acf((const int32_t*)c->d_in.ptr, num_samples, (const sla_hip_acf_job*)c->d_job.ptr, 1, c->fft_size,
                                (const double*)c->d_tw.ptr, (double*)c->d_scratch.ptr, slots, (double*)c->d_rec.ptr,
                                SLA_HIP_ACF_RECORD, NULL) != 0
      || down(rec, c->d_rec.ptr, sizeof(rec)) != 0) {
    return SLAPREDICTOR_APIRESULT_NG;
  }
  rc = slai_ltm_solve(rec, num_taps, pitch_num_samples, ltm_coef);
  return (rc == 0) ? SLAPREDICTOR_APIRESULT_OK : SLAPREDICTOR_APIRESULT_FAILED_TO_CALCULATION;
}

/* ------------------------------------------------------------------ long-term filter, LMS, Rice sums: k_tail with stages switched off */

typedef struct { dbuf_t d_in, d_out, d_job, d_fold; } tailbuf_t;

static void tail_free(tailbuf_t* t) { dfree(&t->d_in); dfree(&t->d_out); dfree(&t->d_job); dfree(&t->d_fold); }

/* one (block, channel) through k_tail; pitch 0 = no long-term stage, skip_lms = no LMS stage */
static int tail_once(tailbuf_t* t, const int32_t* data, uint32_t n, uint32_t pitch, const int32_t* coef, uint32_t ntaps,
                     uint32_t lms_order, uint32_t skip_lms, int32_t* residual, uint64_t* fold)
{
  sla_hip_tail_job job;
  uint32_t k;
  memset(&job, 0, sizeof(job));
  job.blk_off = 0; job.blk_len = n; job.channel = 0; job.pitch = pitch;
  for (k = 0; k < ntaps && k < 5; k++) { job.ltm_coef[k] = coef[k]; }
  if (up(&t->d_in, data, sizeof(int32_t) * n) != 0 || up(&t->d_job, &job, sizeof(job)) != 0
      || dres(&t->d_out, sizeof(int32_t) * n) != 0 || dres(&t->d_fold, sizeof(uint64_t)) != 0
      || sla_hip_launch_tail_stages((const int32_t*)t->d_in.ptr, (int32_t*)t->d_out.ptr, n, (const sla_hip_tail_job*)t->d_job.ptr, 1,
                                    ntaps, lms_order, skip_lms, (uint64_t*)t->d_fold.ptr, NULL) != 0) {
    return -1;
  }
  if (residual != NULL && down(residual, t->d_out.ptr, sizeof(int32_t) * n) != 0) { return -1; }
  if (fold != NULL && down(fold, t->d_fold.ptr, sizeof(uint64_t)) != 0) { return -1; }
  return 0;
}

struct SLALongTermSynthesizer { uint32_t max_taps, max_period; int used; tailbuf_t t; };

struct SLALongTermSynthesizer* SLALongTermSynthesizer_Create(uint32_t max_num_taps, uint32_t max_pitch_period)
{
  struct SLALongTermSynthesizer* l;
  if (!dev_ok() || max_num_taps > SLAI_MAX_TAPS) { return NULL; }
  l = (struct SLALongTermSynthesizer*)calloc(1, sizeof(*l));
  if (l != NULL) { l->max_taps = max_num_taps; l->max_period = max_pitch_period; }
  return l;
}

void SLALongTermSynthesizer_Destroy(struct SLALongTermSynthesizer* l) { if (l != NULL) { tail_free(&l->t); free(l); } }

SLAPredictorApiResult SLALongTermSynthesizer_Reset(struct SLALongTermSynthesizer* l)
{
  if (l == NULL) { return SLAPREDICTOR_APIRESULT_INVALID_ARGUMENT; }
  l->used = 0;
  return SLAPREDICTOR_APIRESULT_OK;
}

SLAPredictorApiResult SLALongTermSynthesizer_PredictInt32(
    struct SLALongTermSynthesizer* l, const int32_t* data, uint32_t num_samples,
    uint32_t pitch_period, const int32_t* ltm_coef, uint32_t num_taps, int32_t* residual)
{
  if (l == NULL || data == NULL || ltm_coef == NULL || residual == NULL) { return SLAPREDICTOR_APIRESULT_INVALID_ARGUMENT; }
  if (pitch_period == 0) { memmove(residual, data, sizeof(int32_t) * num_samples); return SLAPREDICTOR_APIRESULT_OK; }   /* src/SLAPredictor.c:1049-1053 */
  if (num_taps > l->max_taps || !(num_taps & 1u)) { return SLAPREDICTOR_APIRESULT_EXCEED_MAX_ORDER; }
  if (pitch_period < SLAI_LTM_MIN_PITCH || pitch_period > l->max_period || l->used) { return SLAPREDICTOR_APIRESULT_NG; }
  if (num_samples == 0) { return SLAPREDICTOR_APIRESULT_OK; }
  l->used = 1;
  return (tail_once(&l->t, data, num_samples, pitch_period, ltm_coef, num_taps, 8, 1, residual, NULL) == 0)
             ? SLAPREDICTOR_APIRESULT_OK : SLAPREDICTOR_APIRESULT_NG;
}

struct SLALMSFilter { uint32_t max_coef; int used; tailbuf_t t; };

struct SLALMSFilter* SLALMSFilter_Create(uint32_t max_num_coef)
{
  struct SLALMSFilter* f;
  if (!dev_ok()) { return NULL; }
  f = (struct SLALMSFilter*)calloc(1, sizeof(*f));
  if (f != NULL) { f->max_coef = max_num_coef; }
  return f;
}

void SLALMSFilter_Destroy(struct SLALMSFilter* f) { if (f != NULL) { tail_free(&f->t); free(f); } }

SLAPredictorApiResult SLALMSFilter_Reset(struct SLALMSFilter* f)
{
  if (f == NULL) { return SLAPREDICTOR_APIRESULT_INVALID_ARGUMENT; }
  f->used = 0;
  return SLAPREDICTOR_APIRESULT_OK;
}

SLAPredictorApiResult SLALMSFilter_PredictInt32(
    struct SLALMSFilter* f, uint32_t num_coef, const int32_t* data, uint32_t num_samples, int32_t* residual)
{
  const int32_t none[5] = {0, 0, 0, 0, 0};
  if (f == NULL || data == NULL || residual == NULL) { return SLAPREDICTOR_APIRESULT_INVALID_ARGUMENT; }
  if (num_coef > f->max_coef) { return SLAPREDICTOR_APIRESULT_EXCEED_MAX_ORDER; }
  if (!(num_coef == 4 || num_coef == 8 || num_coef == 16 || num_coef == 32) || f->used) { return SLAPREDICTOR_APIRESULT_NG; }
  if (num_samples == 0) { return SLAPREDICTOR_APIRESULT_OK; }
  f->used = 1;
  return (tail_once(&f->t, data, num_samples, 0, none, 1, num_coef, 0, residual, NULL) == 0)
             ? SLAPREDICTOR_APIRESULT_OK : SLAPREDICTOR_APIRESULT_NG;
}

/* ------------------------------------------------------------------ decode side: one block through the synthesis kernels */

/* `data` (n samples) through one stage of sla_decode.hip on a batch of one block: stage 0 = LMS synthesis, 1 = long-term
 * synthesis, 2 = PARCOR synthesis lattice (no de-emphasis), 3 = de-emphasis */
static int synth_once(int stage, const int32_t* data, uint32_t n, int32_t* out, uint32_t lms_order, uint32_t pitch,
                      const int32_t* ltm_coef, uint32_t ntaps, const int32_t* kint, uint32_t order, int32_t previous, uint32_t shift)
{
  dbuf_t d_x = {NULL, 0}, d_blk = {NULL, 0}, d_info = {NULL, 0}, d_chan = {NULL, 0}, d_k = {NULL, 0};
  sla_hip_dec_block blk;
  sla_hip_dec_info info;
  sla_hip_dec_chan chan;
  uint32_t t;
  int rc = -1;
  memset(&blk, 0, sizeof(blk)); memset(&info, 0, sizeof(info)); memset(&chan, 0, sizeof(chan));
  blk.num_samples = n;
  chan.pitch = pitch;
  for (t = 0; t < ntaps && t < 5; t++) { chan.ltm_coef[t] = ltm_coef[t]; }
  if (up(&d_x, data, sizeof(int32_t) * n) != 0 || up(&d_blk, &blk, sizeof(blk)) != 0 || up(&d_info, &info, sizeof(info)) != 0
      || up(&d_chan, &chan, sizeof(chan)) != 0) { goto done; }
  if (stage == 0) {
    rc = sla_hip_launch_dec_lms((int32_t*)d_x.ptr, n, (const sla_hip_dec_block*)d_blk.ptr, (const sla_hip_dec_info*)d_info.ptr, 1, 1, lms_order, NULL);
  } else if (stage == 1) {
    rc = sla_hip_launch_dec_ltm((int32_t*)d_x.ptr, n, (const sla_hip_dec_block*)d_blk.ptr, (const sla_hip_dec_info*)d_info.ptr,
                                (const sla_hip_dec_chan*)d_chan.ptr, 1, 1, ntaps, n, NULL);
  } else if (stage == 2) {
    if (up(&d_k, kint, sizeof(int32_t) * (order + 1)) != 0) { goto done; }
    rc = sla_hip_launch_dec_lattice((int32_t*)d_x.ptr, n, (const sla_hip_dec_block*)d_blk.ptr, (const sla_hip_dec_info*)d_info.ptr, 1, 1,
                                    (const int32_t*)d_k.ptr, order, 0, NULL);
  } else {
    rc = sla_hip_launch_dec_deemphasis((int32_t*)d_x.ptr, n, previous, shift, NULL);
  }
  if (rc == 0) { rc = down(out, d_x.ptr, sizeof(int32_t) * n); }
done:
  dfree(&d_x); dfree(&d_blk); dfree(&d_info); dfree(&d_chan); dfree(&d_k);
  return rc;
}

/* reference src/SLAPredictor.c:610-740 */
SLAPredictorApiResult SLALPCSynthesizer_SynthesizeByParcorCoefInt32(
    struct SLALPCSynthesizer* s, const int32_t* residual, uint32_t num_samples,
    const int32_t* parcor_coef, uint32_t order, int32_t* output)
{
  if (s == NULL || residual == NULL || parcor_coef == NULL || output == NULL) { return SLAPREDICTOR_APIRESULT_INVALID_ARGUMENT; }
  if (order > s->max_order) { return SLAPREDICTOR_APIRESULT_EXCEED_MAX_ORDER; }
  if (s->used) { return SLAPREDICTOR_APIRESULT_NG; }             /* continuing a block is not supported: Reset first */
  if (num_samples == 0) { return SLAPREDICTOR_APIRESULT_OK; }
  s->used = 1;
  if (order == 0) { memmove(output, residual, sizeof(int32_t) * num_samples); return SLAPREDICTOR_APIRESULT_OK; }
  if (order > 255) { return SLAPREDICTOR_APIRESULT_NG; }
  return (synth_once(2, residual, num_samples, output, 0, 0, NULL, 0, parcor_coef, order, 0, 0) == 0)
             ? SLAPREDICTOR_APIRESULT_OK : SLAPREDICTOR_APIRESULT_NG;
}

/* reference src/SLAPredictor.c:1034-1119 */
SLAPredictorApiResult SLALongTermSynthesizer_SynthesizeInt32(
    struct SLALongTermSynthesizer* l, const int32_t* residual, uint32_t num_samples,
    uint32_t pitch_period, const int32_t* ltm_coef, uint32_t num_taps, int32_t* output)
{
  if (l == NULL || residual == NULL || ltm_coef == NULL || output == NULL) { return SLAPREDICTOR_APIRESULT_INVALID_ARGUMENT; }
  if (pitch_period == 0) { memmove(output, residual, sizeof(int32_t) * num_samples); return SLAPREDICTOR_APIRESULT_OK; }
  if (num_taps > l->max_taps || !(num_taps & 1u)) { return SLAPREDICTOR_APIRESULT_EXCEED_MAX_ORDER; }
  if (pitch_period > l->max_period || l->used || num_samples > 16384) { return SLAPREDICTOR_APIRESULT_NG; }
  if (num_samples == 0) { return SLAPREDICTOR_APIRESULT_OK; }
  l->used = 1;
  return (synth_once(1, residual, num_samples, output, 0, pitch_period, ltm_coef, num_taps, NULL, 0, 0, 0) == 0)
             ? SLAPREDICTOR_APIRESULT_OK : SLAPREDICTOR_APIRESULT_NG;
}

/* reference src/SLAPredictor.c:1334-1463 */
SLAPredictorApiResult SLALMSFilter_SynthesizeInt32(
    struct SLALMSFilter* f, uint32_t num_coef, const int32_t* residual, uint32_t num_samples, int32_t* output)
{
  if (f == NULL || residual == NULL || output == NULL) { return SLAPREDICTOR_APIRESULT_INVALID_ARGUMENT; }
  if (num_coef > f->max_coef) { return SLAPREDICTOR_APIRESULT_EXCEED_MAX_ORDER; }
  if (!(num_coef == 4 || num_coef == 8 || num_coef == 16 || num_coef == 32) || f->used) { return SLAPREDICTOR_APIRESULT_NG; }
  if (num_samples == 0) { return SLAPREDICTOR_APIRESULT_OK; }
  f->used = 1;
  return (synth_once(0, residual, num_samples, output, num_coef, 0, NULL, 0, NULL, 0, 0, 0) == 0)
             ? SLAPREDICTOR_APIRESULT_OK : SLAPREDICTOR_APIRESULT_NG;
}

/* ------------------------------------------------------------------ partition search */

struct SLAOptimalBlockPartitionEstimator { uint32_t max_nodes; };

#define OEE_NUM_NODES(n, delta) ((((n) + (delta) - 1) / (delta)) + 1)       /* src/SLAPredictor.c:23-24 */

uint32_t SLAOptimalEncodeEstimator_CalculateMaxNumPartitions(uint32_t max_num_samples, uint32_t delta_num_samples)
{
  return OEE_NUM_NODES(max_num_samples, delta_num_samples);
}

struct SLAOptimalBlockPartitionEstimator* SLAOptimalEncodeEstimator_Create(uint32_t max_num_samples, uint32_t delta_num_samples)
{
  struct SLAOptimalBlockPartitionEstimator* o;
  if (!dev_ok() || delta_num_samples == 0 || max_num_samples == 0) { return NULL; }
  if (OEE_NUM_NODES(max_num_samples, delta_num_samples) > SLAI_MAX_NODES) { return NULL; }
  o = (struct SLAOptimalBlockPartitionEstimator*)calloc(1, sizeof(*o));
  if (o != NULL) { o->max_nodes = OEE_NUM_NODES(max_num_samples, delta_num_samples); }
  return o;
}

void SLAOptimalEncodeEstimator_Destroy(struct SLAOptimalBlockPartitionEstimator* o) { free(o); }

SLAPredictorApiResult SLAOptimalEncodeEstimator_SearchOptimalBlockPartitions(
    struct SLAOptimalBlockPartitionEstimator* o, struct SLALPCCalculator* lpcc,
    const double* const* data, uint32_t num_channels, uint32_t num_samples,
    uint32_t min_num_block_samples, uint32_t delta_num_samples, uint32_t max_num_block_samples,
    uint32_t bits_per_sample, uint32_t parcor_order,
    uint32_t* optimal_num_partitions, uint32_t* optimal_block_partition)
{
  uint32_t nodes, i, j, ch, ncand = 0, count, node, O2 = parcor_order + 2;
  uint32_t* pair;
  sla_hip_lpc_cand* cands;
  double *out, *adj;
  uint32_t path[SLAI_MAX_NODES];
  SLAPredictorApiResult res = SLAPREDICTOR_APIRESULT_FAILED_TO_CALCULATION;
  if (o == NULL || lpcc == NULL || data == NULL || optimal_num_partitions == NULL || optimal_block_partition == NULL) {
    return SLAPREDICTOR_APIRESULT_INVALID_ARGUMENT;
  }
  if (delta_num_samples == 0 || num_samples == 0) { return SLAPREDICTOR_APIRESULT_INVALID_ARGUMENT; }
  nodes = OEE_NUM_NODES(num_samples, delta_num_samples);
  if (nodes > o->max_nodes) { return SLAPREDICTOR_APIRESULT_EXCEED_MAX_ORDER; }
  if (parcor_order > lpcc->max_order) { return SLAPREDICTOR_APIRESULT_FAILED_TO_CALCULATION; }
  if (parcor_order == 0 || parcor_order > 255 || num_samples > PREDICTOR_MAX_SAMPLES) { return SLAPREDICTOR_APIRESULT_NG; }
  pair = (uint32_t*)malloc(sizeof(uint32_t) * nodes * nodes);
  cands = (sla_hip_lpc_cand*)malloc(sizeof(*cands) * nodes * nodes);
  adj = (double*)calloc((size_t)nodes * nodes, sizeof(double));
  out = NULL;
  if (pair == NULL || cands == NULL || adj == NULL) { goto done; }
  /* every (i,j), j > i, whose clipped length is allowed                      src/SLAPredictor.c:1615-1630 */
  for (i = 0; i < nodes; i++) {
    for (j = 0; j < nodes; j++) {
      uint32_t off = i * delta_num_samples, len = (j > i) ? (j - i) * delta_num_samples : 0;
      pair[i * nodes + j] = 0xFFFFFFFFu;
      if (j <= i || off >= num_samples) { continue; }
      if (len > num_samples - off) { len = num_samples - off; }
      if (len < min_num_block_samples || len > max_num_block_samples) { continue; }
      pair[i * nodes + j] = ncand;
      cands[ncand].start = off; cands[ncand].len = len; ncand++;
    }
  }
  for (i = 0; i < nodes * nodes; i++) { adj[i] = SLAI_BIG_WEIGHT; }
  if (ncand > 0) {
    out = (double*)malloc(sizeof(double) * (size_t)ncand * O2);
    if (out == NULL) { goto done; }
    for (i = 0; i < nodes * nodes; i++) { if (pair[i] != 0xFFFFFFFFu) { adj[i] = 0.0; } }
    for (ch = 0; ch < num_channels; ch++) {
      if (data[ch] == NULL || lpc_candidates(lpcc, data[ch], num_samples, parcor_order, cands, ncand, out) != 0) { goto done; }
      for (i = 0; i < nodes * nodes; i++) {
        const uint32_t k = pair[i];
        if (k == 0xFFFFFFFFu) { continue; }
        adj[i] += cands[k].len * slai_code_length(out[(size_t)k * O2], cands[k].len, bits_per_sample, out + (size_t)k * O2 + 1, parcor_order);
      }
    }
    for (i = 0; i < nodes * nodes; i++) {
      if (pair[i] != 0xFFFFFFFFu) { adj[i] += SLAI_EST_BLOCK_HEADER; adj[i] += SLAI_PATH_PENALTY; }
    }
  }
  if (slai_shortest_path(adj, nodes, path) != 0) { goto done; }
  count = 0;
  for (node = nodes - 1; node != 0; node = path[node]) {
    if (path[node] >= node) { goto done; }
    count++;
  }
  node = nodes - 1;
  for (i = 0; i < count; i++) {
    uint32_t off = path[node] * delta_num_samples, len = (node - path[node]) * delta_num_samples;
    if (len > num_samples - off) { len = num_samples - off; }
    optimal_block_partition[count - i - 1] = len;
    node = path[node];
  }
  *optimal_num_partitions = count;
  res = SLAPREDICTOR_APIRESULT_OK;
done:
  free(pair); free(cands); free(adj); free(out);
  return res;
}

/* ------------------------------------------------------------------ pre-emphasis */

struct SLAEmphasisFilter { int32_t prev; };

struct SLAEmphasisFilter* SLAEmphasisFilter_Create(void)
{
  if (!dev_ok()) { return NULL; }
  return (struct SLAEmphasisFilter*)calloc(1, sizeof(struct SLAEmphasisFilter));
}

SLAPredictorApiResult SLAEmphasisFilter_Reset(struct SLAEmphasisFilter* e)
{
  if (e == NULL) { return SLAPREDICTOR_APIRESULT_INVALID_ARGUMENT; }
  e->prev = 0;
  return SLAPREDICTOR_APIRESULT_OK;
}

void SLAEmphasisFilter_Destroy(struct SLAEmphasisFilter* e) { free(e); }

SLAPredictorApiResult SLAEmphasisFilter_PreEmphasisInt32(
    struct SLAEmphasisFilter* e, int32_t* data, uint32_t num_samples, int32_t coef_shift)
{
  dbuf_t in = {NULL, 0}, out = {NULL, 0};
  SLAPredictorApiResult res = SLAPREDICTOR_APIRESULT_NG;
  int32_t last;
  if (e == NULL || data == NULL) { return SLAPREDICTOR_APIRESULT_INVALID_ARGUMENT; }
  if (num_samples == 0) { return SLAPREDICTOR_APIRESULT_OK; }
  if (coef_shift < 1 || coef_shift > 30) { return SLAPREDICTOR_APIRESULT_NG; }
  last = data[num_samples - 1];
  if (up(&in, data, sizeof(int32_t) * num_samples) == 0 && dres(&out, sizeof(int32_t) * num_samples) == 0
      && sla_hip_launch_emphasis_i32((const int32_t*)in.ptr, (int32_t*)out.ptr, num_samples, e->prev, (uint32_t)coef_shift, NULL) == 0
      && down(data, out.ptr, sizeof(int32_t) * num_samples) == 0) {
    e->prev = last;
    res = SLAPREDICTOR_APIRESULT_OK;
  }
  dfree(&in); dfree(&out);
  return res;
}

/* in place; the last output sample is carried to the next call            reference src/SLAPredictor.c:1768-1791 */
SLAPredictorApiResult SLAEmphasisFilter_DeEmphasisInt32(
    struct SLAEmphasisFilter* e, int32_t* data, uint32_t num_samples, int32_t coef_shift)
{
  if (e == NULL || data == NULL) { return SLAPREDICTOR_APIRESULT_INVALID_ARGUMENT; }
  if (num_samples == 0) { return SLAPREDICTOR_APIRESULT_OK; }
  if (coef_shift < 1 || coef_shift > 30) { return SLAPREDICTOR_APIRESULT_NG; }
  if (synth_once(3, data, num_samples, data, 0, 0, NULL, 0, NULL, 0, e->prev, (uint32_t)coef_shift) != 0) { return SLAPREDICTOR_APIRESULT_NG; }
  e->prev = data[num_samples - 1];
  return SLAPREDICTOR_APIRESULT_OK;
}

void SLAEmphasisFilter_PreEmphasisDouble(double* data, uint32_t num_samples, int32_t coef_shift)
{
  dbuf_t in = {NULL, 0}, out = {NULL, 0};
  if (data == NULL || num_samples == 0 || coef_shift < 1 || coef_shift > 30 || !dev_ok()) { return; }
  if (up(&in, data, sizeof(double) * num_samples) == 0 && dres(&out, sizeof(double) * num_samples) == 0
      && sla_hip_launch_emphasis_f64((const double*)in.ptr, (double*)out.ptr, num_samples, (uint32_t)coef_shift, NULL) == 0) {
    (void)down(data, out.ptr, sizeof(double) * num_samples);
  }
  dfree(&in); dfree(&out);
}

/* ------------------------------------------------------------------ coder: initial Rice parameters */

struct SLACoder { uint32_t max_channels, max_params; uint32_t* init; tailbuf_t t; };

struct SLACoder* SLACoder_Create(uint32_t max_num_channels, uint32_t max_num_parameters)
{
  struct SLACoder* c;
  if (!dev_ok() || max_num_channels == 0) { return NULL; }
  c = (struct SLACoder*)calloc(1, sizeof(*c));
  if (c == NULL) { return NULL; }
  c->max_channels = max_num_channels; c->max_params = max_num_parameters;
  c->init = (uint32_t*)calloc(max_num_channels, sizeof(uint32_t));
  if (c->init == NULL) { free(c); return NULL; }
  return c;
}

void SLACoder_Destroy(struct SLACoder* c)
{
  if (c == NULL) { return; }
  tail_free(&c->t); free(c->init); free(c);
}

void SLACoder_CalculateInitialRecursiveRiceParameter(
    struct SLACoder* c, uint32_t num_parameters, const int32_t** data, uint32_t num_channels, uint32_t num_samples)
{
  const int32_t none[5] = {0, 0, 0, 0, 0};
  uint32_t ch;
  if (c == NULL || data == NULL || num_parameters > c->max_params || num_samples == 0) { return; }
  for (ch = 0; ch < num_channels && ch < c->max_channels; ch++) {
    uint64_t fold = 0;
    /* k_tail with both filter stages off passes the samples through and returns the sum of their zig-zag codes */
    if (data[ch] == NULL || tail_once(&c->t, data[ch], num_samples, 0, none, 1, 8, 1, NULL, &fold) != 0) { c->init[ch] = 0; continue; }
    fold /= num_samples;
    c->init[ch] = (fold < 1) ? 1u : (uint32_t)fold;                    /* src/SLACoder.c:378 */
  }
}

uint32_t sla_hip_coder_initial_parameter(const struct SLACoder* c, uint32_t channel)
{
  return (c != NULL && channel < c->max_channels) ? c->init[channel] : 0u;
}
