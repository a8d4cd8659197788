#define _GNU_SOURCE
/*
 * sla_encoder.c -- host orchestration of the MI355X SLA encode path (plain C).
 *
 * Drop-in for the reference encoder object (src/SLAEncoder.c): same public
 * functions and error codes, but instead of walking the file block by block
 * it runs the whole file as a few batched kernel launches:
 *
 *   prepass kernel  -> host: offset_lshift, super-frame table (silence hops)
 *   lpc kernel      -> host: code lengths (libm log) + Dijkstra per super-frame
 *   lpc kernel (windowed, + quantiser), lattice kernel
 *                   -> host: RAW decision, long-term analysis (FFT today on the
 *                      host -- see DESIGN.md "next rows"), tail jobs
 *   tail kernel     -> Rice initial parameters
 *   pack            -> host threads: headers, Rice bodies, CRC16
 *
 * Nothing here computes the hot path on the CPU: without a HIP device
 * SLAEncoder_Create fails.
 */
#include "sla_internal.h"
#include "SLADecoder.h"

#include <math.h>
#include <pthread.h>
#include <sched.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>
#include <unistd.h>

#define STATUS_WAVE_FORMAT   (1u << 0)
#define STATUS_ENCODE_PARAM  (1u << 1)
#define MAX_ANALYSIS_WINDOW  16384u     /* LDS-resident window: 128 KiB of doubles */

typedef struct { void* ptr; size_t cap; } devbuf_t;
typedef struct { void* ptr; size_t cap; } pinbuf_t;

typedef struct {
  uint32_t start, nsmpl, type, bytes;
} blk_t;

/* per (block, channel) results */
typedef struct {
  uint32_t rshift, pitch, rice_init;
  int32_t  ltm_q[SLAI_MAX_TAPS];
} blkch_t;

struct slai_pool;

struct SLAEncoder {
  struct SLAEncoderConfig   cfg;
  struct SLAWaveFormat      wave_format;
  struct SLAEncodeParameter encode_param;
  uint32_t status_flag;
  int      device;
  hipStream_t stream, stream2, stream3;
  int own_copy_streams;
  hipStream_t stream_up, stream_down;            /* descriptor uploads / result downloads: kept off the kernel streams */
  hipEvent_t  ev[2 + 8 * 20];       /* prepass pair + MAX_CHUNKS x EV_PER_CHUNK (checked below) */
  uint32_t chunks;
  uint32_t split[8], split_count, chunk_cut[9];   /* relative chunk sizes (none set: the built-in shares), boundaries of this run */
  int      fuse_lattice;            /* 1: the lattice runs inside k_lpc_blocks (SLA_HIP_LATTICE=fused), 0: separate k_lattice launch */
  int      device_plan;             /* 1: code lengths + Dijkstra on the device where certified (default), 0: host only */
  int      search_exact;            /* 1: tile-sum search where it is provably bit-exact (default), 0: chains only */
  int      exact_bits;              /* log2 of the energy limit in units^2 (53; lowered by tests to force the fallback) */
  double   cert_safety;             /* windows over the limit: safety factor of the partition certificate (64); 0: always rerun them as serial chains */
  int      plan_copy_down;          /* 1: plan results travel on the download stream (debugging; default: search stream) */
  int      chunks_forced;           /* the caller chose the chunk count: no short-file rule */
  uint32_t first_chunk;             /* 1/1000 of the super-frames in chunk 0 (0: built-in shares) */
  int      single_tail;             /* 1: one k_tail launch for all chunks (default) */
  int      alt_streams;             /* option: 0 never, 1 / 2 (default) whenever chunked -- the block stages of odd and even chunks on two streams */
  int      alt_now;                 /* this run's decision */
  int      device_ltm;              /* 1: pitch + taps solved on the device, the block stage runs lattice -> tail without the host (default) */
  int      trace;                   /* host-side timeline on stderr (SLA_HIP_TRACE) */
  int      upload24;                /* 1 (default): pageable input of 17..24 significant bits crosses the bus as three bytes per sample (k_unpack24), see DESIGN 7 */
  int      block_cert;              /* 1 (default): chosen blocks through the any-order autocorrelation where their codes and the RAW decision certify,
                                     * the exact chain kernel for the rest; 0: every block through the exact kernel */
  double   block_cert_safety;       /* safety factor on the first-order bound of that certificate (16) */
  volatile int cert_broken;         /* a block flagged by the certificate was not redone (internal error), or an audited block differed */
  volatile uint32_t audit_ok, audit_bad;   /* option cert_audit: certified pairs the exact kernels found equal / different (last analysis) */
  int      cert_now;                /* this run's block stage takes the certified route */
  int      prelaunch;               /* 1 (default): short files queue the certified block kernels with the searches, sized for the most groups
                                     * there can be and counting on the device (search_launch) */
  int      one_stream;              /* 1: a file that runs as one chunk on device tables keeps search, block stage and tail on ONE stream (measured
                                     * slower; default 0) */
  int      device_expand;           /* 1 (default): the block table of certified partitions is written on the device (k_expand) and the block
                                     * stage launched from two counts, the host's own tables following under the kernels; 0: host tables first */
  int      expand_silence;          /* 1 (default): files / batches WITH silence take device-written block tables too (k_expand reads the
                                     * prepass mask: all-zero blocks of searched super-frames get no group); 0: host tables for them (rounds 2-3) */
  uint32_t expand_seq;              /* sequence number of the last k_expand launch (what the host polls for) */
  uint32_t expanded_chunks;         /* last analysis: pipeline chunks whose block stage was launched from device tables */
  /* Search tables (super-frames, candidate shapes, groups) of the last file WITHOUT silence, host and device copies, kept for
   * the next file of the same shape: they depend on the length and the parameters only, and building + uploading them
   * (40 - 90 us for a ten-minute file) is time the device spends idle behind the prepass.  option "table_cache" */
  int      table_cache, tab_valid;
  uint32_t tab_key[10], tab_cnt[12];
  void*    tab_sf; void* tab_shapes;
  uint32_t winmap_entries;          /* window list entries k_expand's device copy holds */
  uint32_t table_hits;              /* analyses served from the kept tables since the handle was created */
  int      spec_valid; uint32_t spec_or;    /* the last file had no silence / its OR word: the guess for the next one of its shape */
  uint32_t spec_misses;             /* guesses that turned out wrong */
  uint32_t blocks_exact;            /* last analysis: (block, channel) pairs the certificate handed to the exact kernel */
  sla_hip_tuning tune;              /* launcher knobs, named to the launchers at every API entry */
  uint32_t host_planned;            /* last analysis: super-frames whose partition the host had to decide */
  uint32_t fallback_groups;         /* last analysis: search groups that had to take the chain kernel */
  uint32_t tail_launches;           /* last analysis: k_tail launches (1 with the device long-term solve, else one per chunk) */
  slai_fft_plan* fft;
  uint32_t threads;
  struct slai_pool* pool;

  /* device workspace */
  devbuf_t d_pcm, d_res1, d_res2, d_or, d_nz, d_groups, d_cands, d_lpc_out, d_code, d_kint, d_rshift,
           d_winpool, d_chunks, d_jobs, d_fold, d_acf_jobs, d_acf, d_acf_scratch, d_twiddle, d_bgroups, d_bcands, d_blk_out, d_kk, d_pk_jobs, d_pk_blocks, d_pk_hdr, d_image,
           d_xgroups, d_tile_sums, d_fgroups, d_parts, d_nparts, d_pstatus, d_spans, d_cert_flag, d_fb_list, d_fb_count,
           d_sframes, d_winmap, d_run, d_expref;
  int twiddle_ready;
  /* pinned host staging */
  pinbuf_t h_nz, h_groups, h_cands, h_lpc_out, h_code, h_kint, h_rshift, h_chunks, h_jobs, h_fold, h_res, h_pcm, h_acf_jobs, h_acf,
           h_bgroups, h_bcands, h_blk_out, h_pk_jobs, h_pk_blocks, h_pk_hdr, h_xgroups, h_fgroups, h_parts, h_nparts, h_pstatus, h_cert_flag,
           h_sframes, h_winmap, h_counts;
  uint32_t* h_or;
  size_t nz_ones_cap; uint64_t nz_ones_words;    /* h_nz words [0, nz_ones_words) are known to be all ones (h_nz.cap == nz_ones_cap) */
  pinbuf_t h_stage[2]; devbuf_t d_stage[2]; hipEvent_t ev_stage[2];
  hipEvent_t ev_prep;               /* the prepass result has reached the host */
  int      groups_valid;            /* d_groups holds the sliced search groups of the current tables (search_groups_ready) */
  hipEvent_t ev_pack[4];            /* device pack: the blocks of a quarter of the image are written and checksummed */

  /* window pool: tables for every block length seen so far */
  double*   win_host; size_t win_count, win_cap;
  uint32_t* win_len; uint32_t* win_off; uint32_t win_entries, win_entries_cap;
  SLAWindowFunctionType win_type; int win_dirty;

  /* caller-owned residual planes (e.g. torch tensors that feed an RCCL all-gather), optional */
  int32_t* user_res1; int32_t* user_res2; uint64_t user_stride;

  /* one file over several GPUs (sla_hip_shard_*): this handle analyses a range of a longer file whose OR word -- hence
   * offset_lshift and sample unit -- is the whole file's, not the range's.  0: off */
  uint32_t file_or_word;
  int      skip_prepass;            /* with file_or_word: the caller vouches that the file has no all-zero mask word */

  /* batch of files in one pass (sla_hip_encode_batch): the files occupy [seg_start, seg_start + seg_len) of the
   * planes, every start a multiple of SLA_HIP_PREPASS_TILE, all with the same offset_lshift.  nsegs == 0: one file
   * = [0, num_samples). */
  uint32_t nsegs; const uint32_t* seg_start; const uint32_t* seg_len;
  uint32_t batch_lshift, batch_or;
  devbuf_t d_tile_or; pinbuf_t h_tile_or;
  devbuf_t d_binfo; pinbuf_t h_binfo;            /* batch: file starts | lengths | k_batch_scan's three words per file */
  int      batch_silence;                        /* the batch has an all-zero mask word or a silent file tail: whole mask on the host, host tables */
  int      mask_absent;                          /* the mask stayed on the device (nothing in it can make a block SILENT): readers take "all ones" */
  uint32_t* tab_segs; uint32_t tab_nsegs;        /* kept search tables of a batch: the file layout they were built for */

  /* SLAEncoder_EncodeWhole of a long file: pieces cross the bus, are analysed and packed on worker lanes (handles of
   * their own on the same device, one host thread each), so upload, kernels and download of different pieces overlap */
  int      stream_mode;             /* 1 (default): on for files of at least two pieces */
  uint32_t stream_piece;            /* samples (all channels together) per piece */
  uint32_t stream_lanes;            /* worker lanes (1..SLAI_STREAM_LANES) */
  uint32_t batch_lanes;             /* sla_hip_encode_batch: worker lanes a big batch is dealt out to (1: none; default 4) */
  pthread_mutex_t* upload_gate;     /* a lane of a batch: uploads take turns (the parent's mutex), everything behind them overlaps */
  double   batch_stamp[4];          /* last encode_batch_pass: upload begun / done, analysed, delivered (ms, now_ms) */
  struct SLAEncoder* lane[SLAI_STREAM_LANES];
  int      is_lane;
  struct slai_pool* upload_pool;    /* a lane's staging copies of its upload run on the parent's (otherwise idle, larger) pool: uploads take turns */
  int      streamed;                /* the last EncodeWhole ran on the lanes: this handle holds no analysis tables */

  /* last analysis */
  const int32_t* pcm_dev;           /* borrowed or &d_pcm */
  uint64_t stride;
  uint32_t num_samples;
  uint32_t lshift;
  blk_t*   blk; uint32_t num_blocks, blk_cap;
  blkch_t* bc;  size_t bc_cap;
  double*  parcor; int32_t* code; int32_t* kint;   /* [num_blocks*C*(order+1)] */
  uint8_t* parcor_exact;            /* [num_blocks*C]: 1 = parcor[] are the reference's doubles bit for bit, 0 = certified (codes and decisions are the reference's) */
  size_t   coef_cap;
  int      analysed;
  float    timing[12];
  int      wall_clock_khz;          /* rate of the device's constant clock (cached) */
  float    kernel_ms[4];            /* last analysis: on-device execution spans of k_lpc_blocks, k_lattice, k_ltm_acf, k_tail, summed over chunks */
};

#define SPAN_SLOT(e, chunk, kernel) ((unsigned long long*)(e)->d_spans.ptr + ((size_t)(chunk) * 4 + (kernel)) * 2)
#define RES1(e) ((e)->user_res1 != NULL ? (e)->user_res1 : (int32_t*)(e)->d_res1.ptr)
#define RES2(e) ((e)->user_res2 != NULL ? (e)->user_res2 : (int32_t*)(e)->d_res2.ptr)

/* ------------------------------------------------------------------ utilities */

static double now_ms(void)
{
  struct timespec ts;
  clock_gettime(CLOCK_MONOTONIC, &ts);
  return ts.tv_sec * 1e3 + ts.tv_nsec * 1e-6;
}

static int hiprc(hipError_t e) { return (e == hipSuccess) ? 0 : -(int)e; }
#define HIPCHK(call) do { hipError_t e__ = (call); if (e__ != hipSuccess) { return hiprc(e__); } } while (0)
#define RCCHK(call)  do { int r__ = (call); if (r__ != 0) { return r__; } } while (0)

static int dev_reserve(devbuf_t* b, size_t bytes)
{
  if (bytes == 0) { bytes = 16; }
  if (b->cap >= bytes) { return 0; }
  if (b->ptr != NULL) { (void)hipFree(b->ptr); b->ptr = NULL; b->cap = 0; }
  bytes += bytes / 8 + 256;
  HIPCHK(hipMalloc(&b->ptr, bytes));
  b->cap = bytes;
  return 0;
}

static int pin_reserve(pinbuf_t* b, size_t bytes)
{
  if (bytes == 0) { bytes = 16; }
  if (b->cap >= bytes) { return 0; }
  if (b->ptr != NULL) { (void)hipHostFree(b->ptr); b->ptr = NULL; b->cap = 0; }
  bytes += bytes / 8 + 256;
  HIPCHK(hipHostMalloc(&b->ptr, bytes, hipHostMallocDefault));
  b->cap = bytes;
  return 0;
}

/* Persistent host worker pool: run fn(ctx, i) for i in [0,count).  The scalar host stages between
 * kernel launches (code-length logs + Dijkstra, long-term solve, block packing) are short, so paying
 * pthread_create per stage would be a visible fraction of them. */
/* Persistent worker pool.  The pipeline issues many short parallel loops (tens of microseconds of work each)
 * in quick succession, so workers spin on the generation counter for a while before they go to sleep on the
 * condition variable: a futex wake-up costs about as much as one of these loops. */
struct slai_pool {
  pthread_t tid[64];
  uint32_t nworkers;                 /* threads besides the caller */
  pthread_mutex_t mu;
  pthread_cond_t cv_start;
  uint64_t generation;               /* atomic */
  uint32_t sleepers;                 /* under mu */
  int shutdown;
  void (*fn)(void*, uint32_t);
  void* ctx;
  uint32_t count, grain;
  uint32_t next;                     /* atomic */
  uint32_t busy;                     /* atomic: workers that have not finished the current generation */
};

static const uint32_t g_pool_spins = 8000u; /* pause instructions before an idle worker sleeps */
#define POOL_SPINS g_pool_spins

static void pool_drain(struct slai_pool* p)
{
  for (;;) {
    uint32_t lo = __atomic_fetch_add(&p->next, p->grain, __ATOMIC_RELAXED), hi, i;
    if (lo >= p->count) { break; }
    hi = (p->count - lo < p->grain) ? p->count : lo + p->grain;
    for (i = lo; i < hi; i++) { p->fn(p->ctx, i); }
  }
}

static void* pool_worker(void* arg)
{
  struct slai_pool* p = (struct slai_pool*)arg;
  uint64_t seen = 0;
  for (;;) {
    uint32_t spins = 0;
    while (__atomic_load_n(&p->generation, __ATOMIC_ACQUIRE) == seen && spins < POOL_SPINS) { __builtin_ia32_pause(); spins++; }
    if (__atomic_load_n(&p->generation, __ATOMIC_ACQUIRE) == seen) {
      pthread_mutex_lock(&p->mu);
      p->sleepers++;
      while (!p->shutdown && __atomic_load_n(&p->generation, __ATOMIC_ACQUIRE) == seen) { pthread_cond_wait(&p->cv_start, &p->mu); }
      p->sleepers--;
      pthread_mutex_unlock(&p->mu);
    }
    if (p->shutdown) { break; }
    seen = __atomic_load_n(&p->generation, __ATOMIC_ACQUIRE);
    pool_drain(p);
    __atomic_fetch_sub(&p->busy, 1, __ATOMIC_RELEASE);
  }
  return NULL;
}

static struct slai_pool* pool_create(uint32_t threads)
{
  struct slai_pool* p = (struct slai_pool*)calloc(1, sizeof(*p));
  uint32_t t;
  if (p == NULL) { return NULL; }
  pthread_mutex_init(&p->mu, NULL);
  pthread_cond_init(&p->cv_start, NULL);
  if (threads > 64) { threads = 64; }
  for (t = 1; t < threads; t++) {
    if (pthread_create(&p->tid[p->nworkers], NULL, pool_worker, p) == 0) { p->nworkers++; }
  }
  return p;
}

static void pool_destroy(struct slai_pool* p)
{
  uint32_t t;
  if (p == NULL) { return; }
  pthread_mutex_lock(&p->mu);
  p->shutdown = 1;
  __atomic_fetch_add(&p->generation, 1, __ATOMIC_RELEASE);
  pthread_cond_broadcast(&p->cv_start);
  pthread_mutex_unlock(&p->mu);
  for (t = 0; t < p->nworkers; t++) { pthread_join(p->tid[t], NULL); }
  pthread_mutex_destroy(&p->mu); pthread_cond_destroy(&p->cv_start);
  free(p);
}

static void parallel_for(struct slai_pool* p, uint32_t count, void (*fn)(void*, uint32_t), void* ctx)
{
  uint32_t i;
  if (count == 0) { return; }
  if (p == NULL || p->nworkers == 0 || count < 4) {
    for (i = 0; i < count; i++) { fn(ctx, i); }
    return;
  }
  p->fn = fn; p->ctx = ctx; p->count = count;
  __atomic_store_n(&p->next, 0, __ATOMIC_RELAXED);
  p->grain = count / ((p->nworkers + 1) * 8);
  if (p->grain == 0) { p->grain = 1; }
  __atomic_store_n(&p->busy, p->nworkers, __ATOMIC_RELAXED);
  pthread_mutex_lock(&p->mu);                     /* the lock orders the bump against a worker that is about to sleep */
  __atomic_fetch_add(&p->generation, 1, __ATOMIC_RELEASE);
  if (p->sleepers != 0) { pthread_cond_broadcast(&p->cv_start); }
  pthread_mutex_unlock(&p->mu);
  pool_drain(p);
  while (__atomic_load_n(&p->busy, __ATOMIC_ACQUIRE) != 0) { __builtin_ia32_pause(); }
}

/* ------------------------------------------------------------- create / destroy */

static char g_device_name[256] = "";
static int g_trace_on = 0;              /* SLA_HIP_TRACE, read in SLAEncoder_Create */
void SLAEncoder_Destroy(struct SLAEncoder* e);
const char* sla_hip_device_name(void) { return g_device_name; }

struct SLAEncoder* SLAEncoder_Create(const struct SLAEncoderConfig* config)
{
  struct SLAEncoder* e;
  hipDeviceProp_t prop;
  int ndev = 0, i;
  const char* env;
  if (config == NULL) { return NULL; }
  if (config->max_num_channels == 0 || config->max_num_channels > SLAI_MAX_CHANNELS
      || config->max_parcor_order > SLAI_MAX_ORDER || config->max_longterm_order > SLAI_MAX_TAPS) { return NULL; }
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) {
    fprintf(stderr, "libsla_hip: no HIP device available -- the encode path has no CPU fallback\n");
    return NULL;
  }
  if (slai_host_check() != 0) {
    fprintf(stderr, "libsla_hip: this host does not evaluate long double / libm the way the reference's x86-64 glibc build does -- "
                    "bit-identical output cannot be guaranteed, refusing to create an encoder\n");
    return NULL;
  }
  e = (struct SLAEncoder*)calloc(1, sizeof(*e));
  if (e == NULL) { return NULL; }
  e->cfg = *config;
  /* the handle is bound to the device that is current now; every API entry makes it current again */
  if (hipGetDevice(&e->device) != hipSuccess) { goto fail; }
  if (hipGetDeviceProperties(&prop, e->device) == hipSuccess) {
    snprintf(g_device_name, sizeof(g_device_name), "%.160s (%.80s)", prop.name, prop.gcnArchName);
  }
  if (hipStreamCreateWithFlags(&e->stream, hipStreamNonBlocking) != hipSuccess
      || hipStreamCreateWithFlags(&e->stream2, hipStreamNonBlocking) != hipSuccess
      || hipStreamCreateWithFlags(&e->stream3, hipStreamNonBlocking) != hipSuccess
      || hipStreamCreateWithFlags(&e->stream_up, hipStreamNonBlocking) != hipSuccess
      || hipStreamCreateWithFlags(&e->stream_down, hipStreamNonBlocking) != hipSuccess) { goto fail; }
  e->own_copy_streams = 1;
  for (i = 0; i < (int)(sizeof(e->ev) / sizeof(e->ev[0])); i++) { if (hipEventCreate(&e->ev[i]) != hipSuccess) { goto fail; } }
  if (hipEventCreate(&e->ev_stage[0]) != hipSuccess || hipEventCreate(&e->ev_stage[1]) != hipSuccess
      || hipEventCreate(&e->ev_prep) != hipSuccess) { goto fail; }
  for (i = 0; i < 4; i++) { if (hipEventCreate(&e->ev_pack[i]) != hipSuccess) { goto fail; } }
  /* the defaults of every knob; sla_hip_encoder_set_option changes them afterwards.  Nothing but SLA_HIP_TRACE comes from the
   * environment (rounds 1 - 3 read some thirty variables here) */
  /* two chunks of 25 % / 75 % that share one tail launch (the figures are at the chunk cuts in run_pipeline) */
  e->chunks = 2;
  e->alt_streams = 2;
  e->single_tail = 1;
  e->device_ltm = 1;
  e->search_exact = 1; e->exact_bits = 53; e->device_plan = 1; e->cert_safety = 64.0;
  e->block_cert = 1; e->block_cert_safety = 16.0;
  e->table_cache = 1;
  e->prelaunch = 1;
  e->device_expand = 1; e->expand_silence = 1;
  e->upload24 = 1;      /* profiles/r3_pack24_ab_*.json: plain path +13 % (C3) / +15 % (C5) from pageable memory, streamed path +1..3 % */
  e->stream_mode = 1; e->stream_piece = 32u << 20; e->stream_lanes = 6; e->batch_lanes = 4;
  /* measured on C2: the lattice inside k_lpc_blocks costs 0.6 ms per step (9 wave-chunks on the 8 waves of a workgroup
   * that has nothing else left to overlap them with), its own launch 0.27 ms: separate by default */
  e->fuse_lattice = 0;
  e->plan_copy_down = 0;
  env = getenv("SLA_HIP_TRACE");
  e->trace = (env != NULL);
  g_trace_on = e->trace;
  if (hipHostMalloc((void**)&e->h_or, 4096, hipHostMallocDefault) != hipSuccess) { e->h_or = NULL; goto fail; }   /* [0,1] prepass words, [2] rerun counter, +64 B: kernel spans */
  {
    uint32_t fft = 1;
    while (fft < config->max_num_block_samples * 2) { fft <<= 1; }    /* src/SLAEncoder.c:110 */
    if (fft < 8) { fft = 8; }
    e->fft = slai_fft_plan_create(fft);
    if (e->fft == NULL) { goto fail; }
  }
  {
    /* host pool: the CPUs this process may use (a launcher of several ranks on few cores sets option "threads") */
    cpu_set_t set;
    e->threads = (uint32_t)sysconf(_SC_NPROCESSORS_ONLN);
    if (sched_getaffinity(0, sizeof(set), &set) == 0 && CPU_COUNT(&set) > 0 && (uint32_t)CPU_COUNT(&set) < e->threads) { e->threads = (uint32_t)CPU_COUNT(&set); }
  }
  if (e->threads < 1) { e->threads = 1; }
  /* The loops are short: more workers only add wake-up and join time.  Six, and a short spin before an idle worker sleeps:
   * ten workers that spin for most of a millisecond after every loop kept ten cores busy through back-to-back calls, and
   * in a container with a CPU quota (the GPU boxes here: 16 cores' worth per 100 ms) that leaves little room before the
   * whole process is throttled for the rest of the period -- 4 - 10 ms stalls in 1.3 ms steps (tests/tools/step_jitter.py).
   * The steps themselves do not notice: C2 1.32 ms, C3-600 s 2.48, C5-120 s 4.77 with 4, 6 or 10 workers. */
  if (e->threads > 6) { e->threads = 6; }
  e->pool = pool_create(e->threads);
  if (e->pool == NULL) { goto fail; }
  e->win_type = (SLAWindowFunctionType)-1;
  return e;
fail:
  SLAEncoder_Destroy(e);            /* copes with a half-built handle: everything it releases is checked for NULL */
  return NULL;
}

void SLAEncoder_Destroy(struct SLAEncoder* e)
{
  devbuf_t* d[41];
  pinbuf_t* h[29];
  int i;
  if (e == NULL) { return; }
  for (i = 0; i < SLAI_STREAM_LANES; i++) { SLAEncoder_Destroy(e->lane[i]); e->lane[i] = NULL; }
  (void)hipSetDevice(e->device);
  if (e->stream != NULL) { (void)hipStreamSynchronize(e->stream); }
  if (e->stream2 != NULL) { (void)hipStreamSynchronize(e->stream2); }
  if (e->stream3 != NULL) { (void)hipStreamSynchronize(e->stream3); }
  if (e->stream_up != NULL) { (void)hipStreamSynchronize(e->stream_up); }
  if (e->stream_down != NULL) { (void)hipStreamSynchronize(e->stream_down); }
  d[0] = &e->d_pcm; d[1] = &e->d_res1; d[2] = &e->d_res2; d[3] = &e->d_or; d[4] = &e->d_nz; d[5] = &e->d_groups;
  d[6] = &e->d_cands; d[7] = &e->d_lpc_out; d[8] = &e->d_code; d[9] = &e->d_kint; d[10] = &e->d_rshift;
  d[11] = &e->d_winpool; d[12] = &e->d_chunks; d[13] = &e->d_jobs; d[14] = &e->d_fold;
  d[15] = &e->d_acf_jobs; d[16] = &e->d_acf; d[17] = &e->d_acf_scratch; d[18] = &e->d_twiddle;
  d[19] = &e->d_bgroups; d[20] = &e->d_bcands; d[21] = &e->d_blk_out;
  d[22] = &e->d_kk; d[23] = &e->d_pk_jobs; d[24] = &e->d_pk_blocks; d[25] = &e->d_pk_hdr; d[26] = &e->d_image;
  d[27] = &e->d_xgroups; d[28] = &e->d_tile_sums; d[29] = &e->d_fgroups;
  d[30] = &e->d_parts; d[31] = &e->d_nparts; d[32] = &e->d_pstatus; d[33] = &e->d_spans;
  d[34] = &e->d_cert_flag; d[35] = &e->d_fb_list; d[36] = &e->d_fb_count;
  d[37] = &e->d_sframes; d[38] = &e->d_winmap; d[39] = &e->d_run; d[40] = &e->d_expref;
  for (i = 0; i < 41; i++) { if (d[i]->ptr != NULL) { (void)hipFree(d[i]->ptr); } }
  h[0] = &e->h_nz; h[1] = &e->h_groups; h[2] = &e->h_cands; h[3] = &e->h_lpc_out; h[4] = &e->h_code; h[5] = &e->h_kint;
  h[6] = &e->h_rshift; h[7] = &e->h_chunks; h[8] = &e->h_jobs; h[9] = &e->h_fold; h[10] = &e->h_res; h[11] = &e->h_pcm;
  h[12] = &e->h_acf_jobs; h[13] = &e->h_acf; h[14] = &e->h_bgroups; h[15] = &e->h_bcands; h[16] = &e->h_blk_out;
  h[17] = &e->h_pk_jobs; h[18] = &e->h_pk_blocks; h[19] = &e->h_pk_hdr; h[20] = &e->h_xgroups; h[21] = &e->h_fgroups;
  h[22] = &e->h_parts; h[23] = &e->h_nparts; h[24] = &e->h_pstatus; h[25] = &e->h_cert_flag;
  h[26] = &e->h_sframes; h[27] = &e->h_winmap; h[28] = &e->h_counts;
  for (i = 0; i < 29; i++) { if (h[i]->ptr != NULL) { (void)hipHostFree(h[i]->ptr); } }
  if (e->h_or != NULL) { (void)hipHostFree(e->h_or); }
  if (e->d_tile_or.ptr != NULL) { (void)hipFree(e->d_tile_or.ptr); }
  if (e->h_tile_or.ptr != NULL) { (void)hipHostFree(e->h_tile_or.ptr); }
  if (e->d_binfo.ptr != NULL) { (void)hipFree(e->d_binfo.ptr); }
  if (e->h_binfo.ptr != NULL) { (void)hipHostFree(e->h_binfo.ptr); }
  free(e->tab_segs);
  for (i = 0; i < 2; i++) {
    if (e->h_stage[i].ptr != NULL) { (void)hipHostFree(e->h_stage[i].ptr); }
    if (e->d_stage[i].ptr != NULL) { (void)hipFree(e->d_stage[i].ptr); }
    if (e->ev_stage[i] != NULL) { (void)hipEventDestroy(e->ev_stage[i]); }
  }
  for (i = 0; i < (int)(sizeof(e->ev) / sizeof(e->ev[0])); i++) { if (e->ev[i] != NULL) { (void)hipEventDestroy(e->ev[i]); } }
  if (e->ev_prep != NULL) { (void)hipEventDestroy(e->ev_prep); }
  for (i = 0; i < 4; i++) { if (e->ev_pack[i] != NULL) { (void)hipEventDestroy(e->ev_pack[i]); } }
  if (e->own_copy_streams) {
    if (e->stream_up != NULL) { (void)hipStreamDestroy(e->stream_up); }
    if (e->stream_down != NULL) { (void)hipStreamDestroy(e->stream_down); }
  }
  if (e->stream != NULL) { (void)hipStreamDestroy(e->stream); }
  if (e->stream2 != NULL) { (void)hipStreamDestroy(e->stream2); }
  if (e->stream3 != NULL) { (void)hipStreamDestroy(e->stream3); }
  if (e->fft != NULL) { slai_fft_plan_destroy(e->fft); }
  pool_destroy(e->pool);
  free(e->win_host); free(e->win_len); free(e->win_off);
  free(e->tab_sf); free(e->tab_shapes);
  free(e->blk); free(e->bc); free(e->parcor); free(e->code); free(e->kint); free(e->parcor_exact);
  free(e);
}

SLAApiResult SLAEncoder_SetWaveFormat(struct SLAEncoder* e, const struct SLAWaveFormat* wf)
{
  if (e == NULL || wf == NULL) { return SLA_APIRESULT_INVALID_ARGUMENT; }
  if (wf->num_channels > e->cfg.max_num_channels || wf->bit_per_sample > 32) { return SLA_APIRESULT_EXCEED_HANDLE_CAPACITY; }
  e->wave_format = *wf;
  e->status_flag |= STATUS_WAVE_FORMAT;
  e->analysed = 0;
  return SLA_APIRESULT_OK;
}

SLAApiResult SLAEncoder_SetEncodeParameter(struct SLAEncoder* e, const struct SLAEncodeParameter* ep)
{
  if (e == NULL || ep == NULL) { return SLA_APIRESULT_INVALID_ARGUMENT; }
  if (ep->parcor_order > e->cfg.max_parcor_order || ep->longterm_order > e->cfg.max_longterm_order
      || ep->lms_order_per_filter > e->cfg.max_lms_order_per_filter
      || ep->max_num_block_samples > e->cfg.max_num_block_samples
      || ep->max_num_block_samples < SLAI_MIN_BLOCK) { return SLA_APIRESULT_EXCEED_HANDLE_CAPACITY; }
  /* device limits of this implementation: the analysis window lives in LDS */
  if (ep->max_num_block_samples > MAX_ANALYSIS_WINDOW || ep->parcor_order < 1) { return SLA_APIRESULT_EXCEED_HANDLE_CAPACITY; }
  e->encode_param = *ep;
  e->status_flag |= STATUS_ENCODE_PARAM;
  e->analysed = 0;
  return SLA_APIRESULT_OK;
}

SLAApiResult SLAEncoder_EncodeHeader(const struct SLAHeaderInfo* header, uint8_t* data, uint32_t data_size)
{
  return (SLAApiResult)slai_write_header(header, data, data_size);
}

/* ------------------------------------------------------------------ window pool */

static int window_offset(struct SLAEncoder* e, uint32_t n, uint32_t* off)
{
  uint32_t i;
  if (e->win_type != e->encode_param.window_function_type) {
    e->win_entries = 0; e->win_count = 0; e->win_type = e->encode_param.window_function_type; e->win_dirty = 1;
  }
  for (i = 0; i < e->win_entries; i++) { if (e->win_len[i] == n) { *off = e->win_off[i]; return 0; } }
  if (e->win_entries == e->win_entries_cap) {
    e->win_entries_cap = e->win_entries_cap ? e->win_entries_cap * 2 : 16;
    e->win_len = (uint32_t*)realloc(e->win_len, sizeof(uint32_t) * e->win_entries_cap);
    e->win_off = (uint32_t*)realloc(e->win_off, sizeof(uint32_t) * e->win_entries_cap);
  }
  if (e->win_count + n > e->win_cap) {
    e->win_cap = (e->win_count + n) * 2;
    e->win_host = (double*)realloc(e->win_host, sizeof(double) * e->win_cap);
  }
  if (e->win_len == NULL || e->win_off == NULL || e->win_host == NULL) { return SLA_APIRESULT_NG; }
  if (slai_make_window(e->win_type, e->win_host + e->win_count, n) != 0) { return SLA_APIRESULT_INVALID_WINDOWFUNCTION_TYPE; }
  e->win_len[e->win_entries] = n; e->win_off[e->win_entries] = (uint32_t)e->win_count;
  *off = (uint32_t)e->win_count;
  e->win_entries++; e->win_count += n; e->win_dirty = 1;
  return 0;
}

/* ------------------------------------------------------------------ block table */

static int blocks_push(struct SLAEncoder* e, uint32_t start, uint32_t nsmpl, uint32_t type)
{
  if (e->num_blocks == e->blk_cap) {
    e->blk_cap = e->blk_cap ? e->blk_cap * 2 : 1024;
    e->blk = (blk_t*)realloc(e->blk, sizeof(blk_t) * e->blk_cap);
    if (e->blk == NULL) { return SLA_APIRESULT_NG; }
  }
  e->blk[e->num_blocks].start = start; e->blk[e->num_blocks].nsmpl = nsmpl;
  e->blk[e->num_blocks].type = type; e->blk[e->num_blocks].bytes = 0;
  e->num_blocks++;
  return 0;
}

/* ---------------------------------------------------------------- analysis pipeline
 *
 * The file is cut into up to MAX_CHUNKS runs of super-frames that travel through three stages
 *   search : k_lpc over every partition candidate           (stream 1)  -> host: log2 costs + Dijkstra
 *   blocks : k_lpc(windowed)+quantiser, k_lattice, k_ltm_acf (stream 2)  -> host: RAW test + Toeplitz solve
 *   tail   : k_tail                                          (stream 3)  -> Rice initial parameters
 * one chunk behind each other, so the host's scalar work on chunk c overlaps kernels of chunks c+1.. and
 * kernels of different stages co-run on the device (k_lpc is LDS-occupancy bound, k_tail/k_lattice use no LDS).
 * All buffers are sized for the whole file before the first launch: nothing is reallocated in flight. */

#define MAX_CHUNKS 8
#define SPECULATE_MAX_GROUPS 16384u      /* (super-frame, channel) pairs up to which the searches are launched on a guess */
enum { EV_SEARCH_S, EV_SEARCH_E, EV_SEARCH_DONE, EV_LPCB_S, EV_LPCB_E, EV_LAT_E, EV_ACF_S, EV_ACF_E, EV_BLOCK_DONE,
       EV_TAIL_S, EV_TAIL_E, EV_TAIL_DONE, EV_UPLOADED, EV_PLANNED, EV_UPLOADED2, EV_LPC_DOWN, EV_SOLVED, EV_EXPANDED, EV_PER_CHUNK };
typedef char ev_array_holds_every_chunk[(2 + MAX_CHUNKS * EV_PER_CHUNK <= 2 + 8 * 20) ? 1 : -1];

typedef struct { uint32_t start, window, min_blk, shape, slot_base, grp_lo, grp_hi, xg; } sframe_t;
typedef struct { uint32_t window, min_blk, nodes, ncand, cand_first; uint32_t pair[SLAI_MAX_NODES * SLAI_MAX_NODES]; } shape_t;

typedef struct {
  uint32_t sf_lo, sf_hi;          /* super-frames of this chunk                                  */
  uint32_t grp_lo, grp_hi;        /* search work-groups                                          */
  uint32_t slot_lo, slot_hi;      /* search output slots                                         */
  uint32_t xg_lo, xg_hi;          /* exact-search groups: one per (live super-frame, channel)    */
  uint32_t blk_lo, blk_hi;        /* blocks the plan produced                                    */
  uint32_t bg_lo, bg_hi;          /* (non-silent block, channel) groups = k_lpc groups = acf jobs */
  uint32_t lc_lo, lc_hi;          /* lattice chunks                                              */
  uint32_t job_lo, job_hi;        /* tail jobs                                                   */
} chunk_t;

typedef struct {
  sframe_t* sf; uint32_t nsf;
  shape_t* shapes; uint32_t nshapes;
  uint32_t ncands, nsgroups, nslots, max_window, max_cpg, nxg, max_xcands;
  int exact;                                      /* this run searches with tile sums            */
  uint32_t blocks_bound, lchunks_bound;
  uint32_t nbg, nlc, njobs, ltm_done;                      /* running counters of the block stage */
  uint32_t *job_blk, *job_ch, *job_grp, *grp_of_slot;
  uint32_t *parts, *nparts; int* status;          /* per super-frame plan results        */
  chunk_t ck[MAX_CHUNKS]; uint32_t nchunks;
  hipEvent_t* ev;                                 /* [nchunks][EV_PER_CHUNK]             */
  /* device-written block tables (k_expand) */
  int expand;                                     /* this run asks for them                       */
  int silence;                                    /* the file (batch) has silence: k_expand reads the mask */
  int borrowed;                                   /* sf / shapes belong to the encoder's table cache */
  int tab_hit;                                    /* ... and were found there                         */
  int spec;                                       /* the searches are in flight on a guess of the prepass result */
  int prelaunched;                                /* chunk 0's certified block kernels were queued with the searches (device-side count) */
  int clear_in_kernel;                            /* spans, rerun counter and k_expand's numbers are cleared by the first search kernel */
  uint32_t* clear_ptr[3]; uint32_t clear_words[3]; /* ... these words: handed to the first search launch (sla_hip_launch_extra), then NULL */
  int one_stream;                                 /* one chunk on device tables: search, block stage and tail on ONE stream (no
                                                   * cross-queue event waits between them: 10 - 20 us each on the critical path) */
  hipStream_t tail_stream;                        /* the stream the last tail kernel was put on */
  uint32_t or_word;                               /* OR of all input words the search was launched with */
  int trace; double t_begin;                      /* for the timeline when the searches go out from pipeline_prepare */
  uint32_t dev_blk, dev_bg;                       /* blocks / groups numbered by the device so far */
  uint8_t launched[MAX_CHUNKS];                   /* the chunk's block stage runs from them        */
  uint32_t dev_lo[MAX_CHUNKS][4];                 /* blk_lo, blk_hi, bg_lo, bg_hi it was launched with */
} actx_t;

typedef struct {
  struct SLAEncoder* e;
  const actx_t* a;
  uint32_t sf_lo;
  const sla_hip_lpc_cand* cands; const double* out;
} plan_ctx_t;

/* host part of the partition search for one super-frame: edge costs from the device's
 * (r0, PARCOR) per candidate, then the shortest path (reference src/SLAPredictor.c:1615-1692) */
static void plan_one(void* vctx, uint32_t rel)
{
  plan_ctx_t* c = (plan_ctx_t*)vctx;
  const struct SLAEncoder* e = c->e;
  const actx_t* a = c->a;
  const uint32_t idx = c->sf_lo + rel;
  const sframe_t* sf = &a->sf[idx];
  const uint32_t C = e->wave_format.num_channels, order = e->encode_param.parcor_order, O2 = order + 2;
  double adj[SLAI_MAX_NODES * SLAI_MAX_NODES];
  uint32_t path[SLAI_MAX_NODES], i, j, ch, count, node;
  const shape_t* sh;
  a->status[idx] = 0; a->nparts[idx] = 0;
  if (sf->shape == 0xFFFFFFFFu) { return; }
  if (e->device_plan) {
    const uint32_t li = sf->xg / C;
    if (((const uint32_t*)e->h_pstatus.ptr)[li] == 0) {          /* decided on the device, certified */
      const uint32_t np = ((const uint32_t*)e->h_nparts.ptr)[li];
      memcpy(a->parts + (size_t)idx * SLAI_MAX_NODES, (const uint32_t*)e->h_parts.ptr + (size_t)li * SLA_HIP_PLAN_NODES, sizeof(uint32_t) * np);
      a->nparts[idx] = np;
      return;
    }
  }
  sh = &a->shapes[sf->shape];
  for (i = 0; i < sh->nodes; i++) {
    for (j = 0; j < sh->nodes; j++) {
      const uint32_t k = sh->pair[i * sh->nodes + j];
      double est = 0.0;
      adj[i * sh->nodes + j] = SLAI_BIG_WEIGHT;
      if (k == 0xFFFFFFFFu) { continue; }
      for (ch = 0; ch < C; ch++) {
        const double* o = c->out + (size_t)(sf->slot_base + ch * sh->ncand + k) * O2;
        const uint32_t len = c->cands[sh->cand_first + k].len;
        est += len * slai_code_length(o[0], len, e->wave_format.bit_per_sample, o + 1, order);
      }
      est += SLAI_EST_BLOCK_HEADER;
      est += SLAI_PATH_PENALTY;
      adj[i * sh->nodes + j] = est;
    }
  }
  if (slai_shortest_path(adj, sh->nodes, path) != 0) { a->status[idx] = SLA_APIRESULT_FAILED_TO_CALCULATE_COEF; return; }
  count = 0;
  for (node = sh->nodes - 1; node != 0; node = path[node]) {
    if (path[node] >= node) { a->status[idx] = SLA_APIRESULT_FAILED_TO_CALCULATE_COEF; return; }
    count++;
  }
  node = sh->nodes - 1;
  for (i = 0; i < count; i++) {
    uint32_t off = path[node] * SLAI_SEARCH_DELTA, len = (node - path[node]) * SLAI_SEARCH_DELTA;
    if (len > sf->window - off) { len = sf->window - off; }
    a->parts[(size_t)idx * SLAI_MAX_NODES + (count - i - 1)] = len;
    node = path[node];
  }
  a->nparts[idx] = count;
}

typedef struct {
  struct SLAEncoder* e;
  const double* acf;           /* [groups][SLAI_LTM_ACF_HEAD] compact autocorrelation records from k_ltm_acf */
  const uint32_t* job_blk; const uint32_t* job_ch; const uint32_t* job_grp;
  uint32_t job_lo;
} ltm_ctx_t;

/* pitch + taps of one (block, channel) from the device's autocorrelation record */
static void ltm_one(void* vctx, uint32_t rel)
{
  ltm_ctx_t* c = (ltm_ctx_t*)vctx;
  struct SLAEncoder* e = c->e;
  const uint32_t j = c->job_lo + rel;
  const uint32_t C = e->wave_format.num_channels, ntaps = e->encode_param.longterm_order;
  blkch_t* bc = &e->bc[(size_t)c->job_blk[j] * C + c->job_ch[j]];
  double coef[SLAI_MAX_TAPS] = {0, 0, 0, 0, 0};
  uint32_t t;
  const int ret = slai_ltm_solve(c->acf + (size_t)c->job_grp[j] * SLAI_LTM_ACF_HEAD, ntaps, &bc->pitch, coef);
  if (ret != 0 || bc->pitch >= SLAI_LTM_MAX_PERIOD) { bc->pitch = 0; }      /* src/SLAEncoder.c:629-632 */
  for (t = 0; t < ntaps; t++) {
    /* Round(coef * 2^15) << 16, x86 conversion semantics       src/SLAEncoder.c:635-640 */
    const double v = coef[t] * 32768.0;
    const double rv = (v >= 0.0) ? floor(v + 0.5) : -floor(-v + 0.5);
    const int32_t q = (!(rv > -2147483649.0 && rv < 2147483648.0)) ? INT32_MIN : (int32_t)rv;
    bc->ltm_q[t] = (int32_t)((uint32_t)q << 16);
  }
}

static void actx_free(actx_t* a)
{
  if (!a->borrowed) { free(a->sf); free(a->shapes); }
  free(a->job_blk); free(a->job_ch); free(a->job_grp); free(a->grp_of_slot);
  free(a->parts); free(a->nparts); free(a->status);
}

/* prepass + whole-file tables: offset_lshift, super-frames, candidate shapes, search groups */
static double g_trace_t0;
#define PTRACE(label) do { if (g_trace_on) { fprintf(stderr, "[sla_hip]   prepare +%7.3f ms  %s\n", now_ms() - g_trace_t0, (label)); } } while (0)

/* whole-file tables: super-frames (hop over silence runs), candidate shapes, search groups.  nz == NULL: no
 * sample is silent (the speculative pass that runs while the prepass is still on the device) */
static int build_tables(struct SLAEncoder* e, actx_t* a, const uint64_t* nz)
{
  const uint32_t C = e->wave_format.num_channels, bps = e->wave_format.bit_per_sample;
  const uint32_t order = e->encode_param.parcor_order, O1 = order + 1;
  const uint32_t n = e->num_samples, maxb = e->encode_param.max_num_block_samples;
  extern uint32_t sla_hip_lattice_chunk_samples(uint32_t order);
  const uint32_t chunk_samples = sla_hip_lattice_chunk_samples(order);
  sla_hip_lpc_cand* cands;
  uint32_t sf_cap = 0, shapes_cap = 8, pos, i, sg;
  const uint32_t nsegs = e->nsegs ? e->nsegs : 1u;

  /* super-frame table (sequential hop over silence runs)        src/SLAEncoder.c:846-869, 392-408
   * -- per file: a batch restarts the hop at every file's first sample */
  a->shapes = (shape_t*)malloc(sizeof(shape_t) * shapes_cap);
  if (a->shapes == NULL) { return SLA_APIRESULT_NG; }
  for (sg = 0; sg < nsegs; sg++) {
  const uint32_t seg_lo = e->nsegs ? e->seg_start[sg] : 0u, seg_hi = e->nsegs ? seg_lo + e->seg_len[sg] : n;
  for (pos = seg_lo; pos < seg_hi;) {
    const uint32_t remain = seg_hi - pos;
    const uint32_t window = (maxb < remain) ? maxb : remain;
    const uint32_t min_blk = (SLAI_MIN_BLOCK < remain) ? SLAI_MIN_BLOCK : remain;
    const uint32_t run = slai_zero_run(nz, pos, window);
    sframe_t* f;
    uint32_t s;
    if (a->nsf == sf_cap) {
      sf_cap = sf_cap ? sf_cap * 2 : 1024;
      a->sf = (sframe_t*)realloc(a->sf, sizeof(sframe_t) * sf_cap);
      if (a->sf == NULL) { return SLA_APIRESULT_NG; }
    }
    f = &a->sf[a->nsf++];
    f->start = pos; f->window = window; f->min_blk = min_blk; f->slot_base = 0; f->grp_lo = f->grp_hi = 0;
    if (run >= min_blk) {
      f->shape = 0xFFFFFFFFu;           /* one SILENT block of `run` samples, no search */
      f->window = run;
      pos += run;
      a->blocks_bound += 1;
    } else {
      for (s = 0; s < a->nshapes; s++) { if (a->shapes[s].window == window && a->shapes[s].min_blk == min_blk) { break; } }
      if (s == a->nshapes) {
        if (a->nshapes == shapes_cap) {
          shapes_cap *= 2;
          a->shapes = (shape_t*)realloc(a->shapes, sizeof(shape_t) * shapes_cap);
          if (a->shapes == NULL) { return SLA_APIRESULT_NG; }
        }
        a->shapes[s].window = window; a->shapes[s].min_blk = min_blk;
        a->shapes[s].nodes = (window + SLAI_SEARCH_DELTA - 1) / SLAI_SEARCH_DELTA + 1;
        a->shapes[s].ncand = 0; a->shapes[s].cand_first = 0;
        a->nshapes++;
      }
      f->shape = s;
      pos += window;
      a->blocks_bound += window / min_blk + 1;
      a->lchunks_bound += C * (window / chunk_samples + window / min_blk + 2);
    }
  }
  }

  PTRACE("super-frame table");
  /* candidate table per shape: every (i,j) whose clipped length is allowed   src/SLAPredictor.c:1615-1630 */
  {
    uint32_t total = 1;
    for (i = 0; i < a->nshapes; i++) { total += a->shapes[i].nodes * a->shapes[i].nodes; }
    RCCHK(pin_reserve(&e->h_cands, sizeof(sla_hip_lpc_cand) * total));
  }
  cands = (sla_hip_lpc_cand*)e->h_cands.ptr;
  for (i = 0; i < a->nshapes; i++) {
    shape_t* sh = &a->shapes[i];
    uint32_t p, q, woff;
    sh->cand_first = a->ncands;
    for (p = 0; p < sh->nodes; p++) {
      for (q = 0; q < sh->nodes; q++) {
        uint32_t off = p * SLAI_SEARCH_DELTA, len = (q > p) ? (q - p) * SLAI_SEARCH_DELTA : 0;
        sh->pair[p * sh->nodes + q] = 0xFFFFFFFFu;
        if (q <= p) { continue; }
        if (len > sh->window - off) { len = sh->window - off; }
        if (len < sh->min_blk || len > sh->window) { continue; }
        sh->pair[p * sh->nodes + q] = sh->ncand;
        cands[a->ncands].start = off; cands[a->ncands].len = len;
        a->ncands++; sh->ncand++;
        RCCHK(window_offset(e, len, &woff));      /* every block length the plan can choose gets its table now */
      }
    }
  }

  PTRACE("candidate + window tables");
  /* search groups: (super-frame, channel), candidates sliced so that window + r[] fits the LDS budget */
  {
    uint32_t total_groups = 1;
    for (i = 0; i < a->nsf; i++) {
      const shape_t* sh;
      size_t room; uint32_t cpg;
      if (a->sf[i].shape == 0xFFFFFFFFu) { continue; }
      sh = &a->shapes[a->sf[i].shape];
      room = (SLA_HIP_LDS_BUDGET / 8 > sh->window) ? (SLA_HIP_LDS_BUDGET / 8 - sh->window) : 0;
      cpg = (uint32_t)(room / O1);
      if (cpg == 0) { return SLA_APIRESULT_EXCEED_HANDLE_CAPACITY; }
      if (cpg > sh->ncand) { cpg = sh->ncand; }
      total_groups += C * ((sh->ncand + cpg - 1) / cpg);
    }
    RCCHK(pin_reserve(&e->h_groups, sizeof(sla_hip_lpc_group) * total_groups));
    RCCHK(pin_reserve(&e->h_xgroups, sizeof(sla_hip_lpc_group) * ((size_t)a->nsf * C + 1)));
  }
  /* The sliced groups are what the serial-chain kernels read: the search without tile sums, and the rerun of what the tile sums
   * could not certify -- the exception.  Here they are only counted; search_groups_ready() writes and uploads them for whoever
   * needs them (an hour of stereo: 84 k descriptors, 3.4 MB, built and sent for nothing on every file whose tables are new). */
  e->groups_valid = 0;
  a->max_window = 1; a->max_cpg = 1; a->max_xcands = 1;
  /* candidates start on multiples of SLAI_SEARCH_DELTA and end on one or with the window: tile aligned */
  a->exact = (e->search_exact && SLAI_SEARCH_DELTA == SLA_HIP_XTILE && maxb <= SLA_HIP_XTILE * SLA_HIP_XTILES
              && sla_hip_search_exact_lags(order) != 0 && e->h_or[0] != 0);
  for (i = 0; i < a->nsf; i++) {
    sframe_t* f = &a->sf[i];
    const shape_t* sh;
    uint32_t cpg, ch, first;
    f->grp_lo = f->grp_hi = a->nsgroups;
    if (f->shape == 0xFFFFFFFFu) { continue; }
    sh = &a->shapes[f->shape];
    cpg = (uint32_t)((SLA_HIP_LDS_BUDGET / 8 - sh->window) / O1);
    if (cpg > sh->ncand) { cpg = sh->ncand; }
    f->slot_base = a->nslots;
    for (ch = 0; ch < C; ch++) {
      for (first = 0; first < sh->ncand; first += cpg) {
        const uint32_t count = (sh->ncand - first < cpg) ? (sh->ncand - first) : cpg;
        a->nsgroups++;
        if (count > a->max_cpg) { a->max_cpg = count; }
      }
    }
    f->grp_hi = a->nsgroups;
    f->xg = a->nxg;
    if (sh->ncand > a->max_xcands) { a->max_xcands = sh->ncand; }
    for (ch = 0; ch < C; ch++) {          /* the same work as one group per channel with every candidate */
      sla_hip_lpc_group* g = (sla_hip_lpc_group*)e->h_xgroups.ptr + a->nxg++;
      g->pcm_off = f->start; g->num_samples = sh->window; g->channel = ch;
      g->win_off = SLA_HIP_NO_WINDOW; g->int_shift = 32 - bps;
      g->cand_first = sh->cand_first; g->cand_count = sh->ncand;
      g->slot_first = a->nslots + ch * sh->ncand; g->pad_ = 0;
    }
    a->nslots += C * sh->ncand;
    if (sh->window > a->max_window) { a->max_window = sh->window; }
  }
  return 0;
}

static int pipeline_reserve(struct SLAEncoder* e, const actx_t* a);

/* the sliced search groups of the current tables (counted by build_tables), written and sent to the device the first time a
 * serial-chain kernel is about to read them; in order on the search stream */
static int search_groups_ready(struct SLAEncoder* e, const actx_t* a)
{
  const uint32_t C = e->wave_format.num_channels, bps = e->wave_format.bit_per_sample;
  const uint32_t O1 = e->encode_param.parcor_order + 1;
  sla_hip_lpc_group* groups = (sla_hip_lpc_group*)e->h_groups.ptr;
  uint32_t i;
  if (e->groups_valid || a->nsgroups == 0) { return 0; }
  if (groups == NULL || e->d_groups.ptr == NULL) { return SLA_APIRESULT_NG; }
  for (i = 0; i < a->nsf; i++) {
    const sframe_t* f = &a->sf[i];
    const shape_t* sh;
    uint32_t cpg, ch, first, g = f->grp_lo;
    if (f->shape == 0xFFFFFFFFu) { continue; }
    sh = &a->shapes[f->shape];
    cpg = (uint32_t)((SLA_HIP_LDS_BUDGET / 8 - sh->window) / O1);
    if (cpg > sh->ncand) { cpg = sh->ncand; }
    for (ch = 0; ch < C; ch++) {
      for (first = 0; first < sh->ncand; first += cpg) {
        sla_hip_lpc_group* gr = &groups[g++];
        gr->pcm_off = f->start; gr->num_samples = sh->window; gr->channel = ch;
        gr->win_off = SLA_HIP_NO_WINDOW; gr->int_shift = 32 - bps;
        gr->cand_first = sh->cand_first + first;
        gr->cand_count = (sh->ncand - first < cpg) ? (sh->ncand - first) : cpg;
        gr->slot_first = f->slot_base + ch * sh->ncand + first; gr->pad_ = 0;
      }
    }
    if (g != f->grp_hi) { return SLA_APIRESULT_NG; }
  }
  HIPCHK(hipMemcpyAsync(e->d_groups.ptr, groups, sizeof(sla_hip_lpc_group) * a->nsgroups, hipMemcpyHostToDevice, e->stream));
  HIPCHK(hipStreamSynchronize(e->stream));            /* (h_groups may be rewritten by the next file's tables) */
  e->groups_valid = 1;
  return 0;
}

/* candidate and group tables of the partition search -> device (in order on the search stream) */
/* In order on the search stream, behind the prepass.  Sending them beside the prepass on another stream, the search stream waiting
 * for an event, was measured (round 4, tests/tools/nocache_ab.sh): the typical C3 step without kept tables got another 0.15 ms
 * shorter, but in one step of three a later launch call of the same analysis then blocked for milliseconds (the block stage went
 * out 7 ms late, the step took 12.6 instead of 9.5 ms) -- on the upload stream and on the block-stage stream alike; not
 * understood, not kept. */
static int upload_search_tables(struct SLAEncoder* e, const actx_t* a)
{
  const hipStream_t us = e->stream;
  if (a->ncands > 0) {
    HIPCHK(hipMemcpyAsync(e->d_cands.ptr, e->h_cands.ptr, sizeof(sla_hip_lpc_cand) * a->ncands, hipMemcpyHostToDevice, us));
  }
  if (a->nxg > 0) {
    HIPCHK(hipMemcpyAsync(e->d_xgroups.ptr, e->h_xgroups.ptr, sizeof(sla_hip_lpc_group) * a->nxg, hipMemcpyHostToDevice, us));
  }
  if (e->device_expand && e->device_plan && a->nsf > 0 && e->win_entries > 0) {
    /* what k_expand reads: the super-frames in file order and the window offset of every block length */
    const uint32_t C = e->wave_format.num_channels;
    sla_hip_superframe* hs; uint32_t* hw; uint32_t i;
    RCCHK(pin_reserve(&e->h_sframes, sizeof(sla_hip_superframe) * a->nsf));
    RCCHK(dev_reserve(&e->d_sframes, sizeof(sla_hip_superframe) * a->nsf));
    RCCHK(pin_reserve(&e->h_winmap, sizeof(uint32_t) * 2 * e->win_entries));
    RCCHK(dev_reserve(&e->d_winmap, sizeof(uint32_t) * 2 * e->win_entries));
    RCCHK(dev_reserve(&e->d_run, 16));
    RCCHK(dev_reserve(&e->d_expref, sizeof(uint32_t) * 2 * ((size_t)a->nsf + 1)));
    RCCHK(pin_reserve(&e->h_counts, sizeof(uint32_t) * 4 * MAX_CHUNKS));
    hs = (sla_hip_superframe*)e->h_sframes.ptr; hw = (uint32_t*)e->h_winmap.ptr;
    for (i = 0; i < a->nsf; i++) {
      hs[i].start = a->sf[i].start; hs[i].window = a->sf[i].window; hs[i].pad_ = 0;
      hs[i].live = (a->sf[i].shape == 0xFFFFFFFFu) ? SLA_HIP_NOT_LIVE : a->sf[i].xg / C;
    }
    for (i = 0; i < e->win_entries; i++) { hw[i] = e->win_len[i]; hw[e->win_entries + i] = e->win_off[i]; }
    HIPCHK(hipMemcpyAsync(e->d_sframes.ptr, hs, sizeof(sla_hip_superframe) * a->nsf, hipMemcpyHostToDevice, us));
    HIPCHK(hipMemcpyAsync(e->d_winmap.ptr, hw, sizeof(uint32_t) * 2 * e->win_entries, hipMemcpyHostToDevice, us));
    e->winmap_entries = e->win_entries;
  }
  return 0;
}

/* tables of a file without silence: from the cache when the last such file had the same shape, else built, uploaded, kept */
static int tables_no_silence(struct SLAEncoder* e, actx_t* a)
{
  uint32_t key[10];
  key[0] = e->num_samples; key[1] = e->wave_format.num_channels; key[2] = e->wave_format.bit_per_sample;
  key[3] = e->encode_param.parcor_order; key[4] = e->encode_param.max_num_block_samples;
  key[5] = (uint32_t)e->encode_param.window_function_type; key[6] = (uint32_t)(e->device_expand && e->device_plan);
  key[7] = e->win_entries; key[8] = (uint32_t)e->win_count; key[9] = 1u;
  if (e->table_cache && e->tab_valid && !e->win_dirty && memcmp(key, e->tab_key, sizeof(key)) == 0) {
    const uint32_t* c = e->tab_cnt;
    a->sf = (sframe_t*)e->tab_sf; a->shapes = (shape_t*)e->tab_shapes; a->borrowed = 1; a->tab_hit = 1;
    a->nsf = c[0]; a->nshapes = c[1]; a->ncands = c[2]; a->nsgroups = c[3]; a->nslots = c[4]; a->max_window = c[5];
    a->max_cpg = c[6]; a->nxg = c[7]; a->max_xcands = c[8]; a->blocks_bound = c[9]; a->lchunks_bound = c[10];
    e->table_hits++;
    return pipeline_reserve(e, a);
  }
  e->tab_valid = 0;
  free(e->tab_sf); free(e->tab_shapes); e->tab_sf = NULL; e->tab_shapes = NULL;
  RCCHK(build_tables(e, a, NULL));
  RCCHK(pipeline_reserve(e, a));
  RCCHK(upload_search_tables(e, a));
  if (e->table_cache) {
    uint32_t* c = e->tab_cnt;
    /* (the window list may have grown while the tables were built: the key holds what it is now) */
    key[7] = e->win_entries; key[8] = (uint32_t)e->win_count;
    c[0] = a->nsf; c[1] = a->nshapes; c[2] = a->ncands; c[3] = a->nsgroups; c[4] = a->nslots; c[5] = a->max_window;
    c[6] = a->max_cpg; c[7] = a->nxg; c[8] = a->max_xcands; c[9] = a->blocks_bound; c[10] = a->lchunks_bound;
    memcpy(e->tab_key, key, sizeof(key));
    e->tab_sf = a->sf; e->tab_shapes = a->shapes; a->borrowed = 1;
    e->tab_valid = 1;
  }
  return 0;
}

/* tables of a batch without silence: functions of the files' positions and lengths (and the parameters) only -- kept for the
 * next batch of the same layout (a directory of equally long clips, a benchmark loop), like a single file's */
static int tables_batch_no_silence(struct SLAEncoder* e, actx_t* a)
{
  uint32_t key[10];
  const size_t seg_bytes = sizeof(uint32_t) * 2 * (size_t)e->nsegs;
  key[0] = e->num_samples; key[1] = e->wave_format.num_channels; key[2] = e->wave_format.bit_per_sample;
  key[3] = e->encode_param.parcor_order; key[4] = e->encode_param.max_num_block_samples;
  key[5] = (uint32_t)e->encode_param.window_function_type; key[6] = (uint32_t)(e->device_expand && e->device_plan);
  key[7] = e->win_entries; key[8] = (uint32_t)e->win_count; key[9] = 2u;
  if (e->table_cache && e->tab_valid && !e->win_dirty && memcmp(key, e->tab_key, sizeof(key)) == 0 && e->tab_nsegs == e->nsegs
      && e->tab_segs != NULL && memcmp(e->tab_segs, e->seg_start, seg_bytes / 2) == 0
      && memcmp(e->tab_segs + e->nsegs, e->seg_len, seg_bytes / 2) == 0) {
    const uint32_t* c = e->tab_cnt;
    a->sf = (sframe_t*)e->tab_sf; a->shapes = (shape_t*)e->tab_shapes; a->borrowed = 1; a->tab_hit = 1;
    a->nsf = c[0]; a->nshapes = c[1]; a->ncands = c[2]; a->nsgroups = c[3]; a->nslots = c[4]; a->max_window = c[5];
    a->max_cpg = c[6]; a->nxg = c[7]; a->max_xcands = c[8]; a->blocks_bound = c[9]; a->lchunks_bound = c[10];
    e->table_hits++;
    return pipeline_reserve(e, a);
  }
  e->tab_valid = 0;
  free(e->tab_sf); free(e->tab_shapes); e->tab_sf = NULL; e->tab_shapes = NULL;
  RCCHK(build_tables(e, a, NULL));                         /* no mask: nothing is silent */
  RCCHK(pipeline_reserve(e, a));
  RCCHK(upload_search_tables(e, a));
  if (e->table_cache) {
    uint32_t* c = e->tab_cnt;
    uint32_t* segs = (uint32_t*)realloc(e->tab_segs, seg_bytes ? seg_bytes : 4);
    if (segs == NULL) { return 0; }                        /* (not kept: the tables stay the analysis context's own) */
    e->tab_segs = segs; e->tab_nsegs = e->nsegs;
    memcpy(segs, e->seg_start, seg_bytes / 2); memcpy(segs + e->nsegs, e->seg_len, seg_bytes / 2);
    key[7] = e->win_entries; key[8] = (uint32_t)e->win_count;
    c[0] = a->nsf; c[1] = a->nshapes; c[2] = a->ncands; c[3] = a->nsgroups; c[4] = a->nslots; c[5] = a->max_window;
    c[6] = a->max_cpg; c[7] = a->nxg; c[8] = a->max_xcands; c[9] = a->blocks_bound; c[10] = a->lchunks_bound;
    memcpy(e->tab_key, key, sizeof(key));
    e->tab_sf = a->sf; e->tab_shapes = a->shapes; a->borrowed = 1;
    e->tab_valid = 1;
  }
  return 0;
}

/* SLA_HIP_TRACE=1: host-side timeline of one analysis on stderr (ms since the start of run_pipeline) */
#define TRACE(label, c) do { if (trace) { fprintf(stderr, "[sla_hip] %8.3f ms  %s %d\n", now_ms() - t_begin, (label), (int)(c)); } } while (0)
static int launch_searches(struct SLAEncoder* e, actx_t* a, int preset_blocks, int trace, double t_begin);

/* offset_lshift = bps - (32 - ntz(OR of all words))          src/SLAEncoder.c:425-455 */
static int lshift_of(uint32_t mask, uint32_t bps, uint32_t* lshift)
{
  *lshift = 0;
  if (mask != 0) {
    const uint32_t ntz = (uint32_t)__builtin_ctz(mask);
    if (bps < 32 - ntz) { return SLA_APIRESULT_INVALID_ARGUMENT; }   /* samples wider than declared */
    *lshift = bps - (32 - ntz);
    if (*lshift >= bps) { return SLA_APIRESULT_INVALID_ARGUMENT; }
  }
  return 0;
}

/* routes that depend on what the prepass found (or is guessed to find): tile-sum search, device-written block tables */
static void decide_routes(struct SLAEncoder* e, actx_t* a, uint32_t or_word, int silence)
{
  const uint32_t order = e->encode_param.parcor_order, maxb = e->encode_param.max_num_block_samples;
  a->or_word = or_word;
  a->exact = (e->search_exact && SLAI_SEARCH_DELTA == SLA_HIP_XTILE && maxb <= SLA_HIP_XTILE * SLA_HIP_XTILES
              && sla_hip_search_exact_lags(order) != 0 && or_word != 0);
  /* Block tables on the device (k_expand).  Where the mask has no all-zero word (or the caller vouched for that) no run of
   * zeros reaches SLA's minimum block length and no block inside a searched super-frame can be SILENT; whole SILENT
   * super-frames are in the table the kernel reads.  With silence (round 4) the kernel reads the device's copy of the mask
   * and leaves the all-zero blocks of searched super-frames without groups, as the host's walk does (plan_chunk, which
   * still runs -- later, under the block kernels -- and must arrive at the same numbers: blocks_launch mode 2).  A batch of
   * files is one super-frame table like any other (the hop restarts at every file). */
  a->silence = silence;
  a->expand = (e->device_expand && e->device_plan && (!silence || (e->expand_silence && e->d_nz.ptr != NULL)) && a->nsf > 0
               && e->winmap_entries > 0 && e->winmap_entries <= 256 && a->max_window <= MAX_ANALYSIS_WINDOW
               && e->d_sframes.ptr != NULL && e->h_counts.ptr != NULL);
}

static int pipeline_prepare(struct SLAEncoder* e, actx_t* a)
{
  const uint32_t C = e->wave_format.num_channels, bps = e->wave_format.bit_per_sample;
  const uint32_t ms = (e->encode_param.ch_process_method == SLA_CHPROCESSMETHOD_STEREO_MS);
  const uint32_t n = e->num_samples, maxb = e->encode_param.max_num_block_samples;
  const uint64_t nwords = ((uint64_t)n + 63) / 64;
  const uint64_t* nz;
  int rebuild = 0;

  g_trace_t0 = now_ms();
  if (e->nsegs != 0) {
    /* batch: sla_hip_encode_batch ran the prepass over all files, the whole mask is on the host and the files of
     * this pass share one offset_lshift */
    e->lshift = e->batch_lshift; e->h_or[0] = e->batch_or;
    if (!e->batch_silence) {
      /* k_batch_scan found nothing that could make a block SILENT: the mask stayed on the device, the tables are the plain
       * hop of every file (kept for the next batch of this layout), the block tables can be written on the device */
      e->mask_absent = 1;
      RCCHK(tables_batch_no_silence(e, a));
      decide_routes(e, a, e->h_or[0], 0);
      return 0;
    }
    e->mask_absent = 0;
    e->tab_valid = 0;                                     /* the batch's tables take the place of the kept ones */
    RCCHK(build_tables(e, a, (const uint64_t*)e->h_nz.ptr));
    RCCHK(pipeline_reserve(e, a));
    RCCHK(upload_search_tables(e, a));
    decide_routes(e, a, e->h_or[0], 1);
    return 0;
  }
  e->mask_absent = 0;
  RCCHK(dev_reserve(&e->d_or, 64));
  RCCHK(dev_reserve(&e->d_nz, (size_t)(nwords + 2) * 8));
  RCCHK(pin_reserve(&e->h_nz, (size_t)(nwords + 2) * 8));
  HIPCHK(hipEventRecord(e->ev[0], e->stream));
  /* (one exception keeps the prepass: a last super-frame of fewer than 127 samples can be all zero -- and then is a SILENT
   * block, its minimum length being what is left of the file, src/SLAEncoder.c:401-407 -- without containing an aligned
   * all-zero 64-sample word for the count to see; without silence the super-frames sit on multiples of the block size) */
  if (e->skip_prepass && e->file_or_word != 0 && (n % maxb == 0 || n % maxb >= 127)) {
    /* sla_hip_shard_analyze_no_silence: the caller's scan of the whole file counted no all-zero mask word, and it knows
     * the file's OR word -- nothing the prepass could add: no super-frame of this range can be silent */
    HIPCHK(hipEventRecord(e->ev[1], e->stream));
    e->h_or[0] = e->file_or_word; e->h_or[1] = 0;
    if (e->nz_ones_cap != e->h_nz.cap) { e->nz_ones_cap = e->h_nz.cap; e->nz_ones_words = 0; }
    if (e->nz_ones_words < nwords) { memset((uint64_t*)e->h_nz.ptr + e->nz_ones_words, 0xFF, (size_t)(nwords - e->nz_ones_words) * 8); }
    e->nz_ones_words = nwords;
    RCCHK(tables_no_silence(e, a));
  } else {
  RCCHK(sla_hip_launch_prepass(e->pcm_dev, e->stride, C, n, bps, ms, (uint32_t*)e->d_or.ptr, (uint64_t*)e->d_nz.ptr, e->stream));
  HIPCHK(hipEventRecord(e->ev[1], e->stream));
  {
    /* Without an all-zero mask word there is no silence to find, except at the very end of the file where the
     * minimum block length shrinks with what is left (src/SLAEncoder.c:846-869): only the last words of the mask
     * are fetched then, everything before them counts as "not silent" (a run of zeros shorter than the minimum
     * block length never changes a decision). */
    const uint64_t tail_words = (nwords < SLAI_MIN_BLOCK / 64 + 2) ? nwords : (SLAI_MIN_BLOCK / 64 + 2);
    const uint64_t head_words = nwords - tail_words;
    HIPCHK(hipMemcpyAsync(e->h_or, e->d_or.ptr, 8, hipMemcpyDeviceToHost, e->stream));
    HIPCHK(hipMemcpyAsync((uint64_t*)e->h_nz.ptr + head_words, (uint64_t*)e->d_nz.ptr + head_words, (size_t)tail_words * 8, hipMemcpyDeviceToHost, e->stream));
    HIPCHK(hipEventRecord(e->ev_prep, e->stream));
    /* while the prepass runs: the tables of a file without silence (the usual case; checked below), every buffer of
     * the pipeline, and the tables on their way to the device behind the prepass -- so that the first search kernel
     * can follow the host's look at the prepass result without further copies in between */
    RCCHK(tables_no_silence(e, a));
    /* The kept tables are a file like the last one: when that one had no silence, guess that this one has none either and
     * the same OR word, and put the searches on the stream right behind the prepass instead of behind the host's look at
     * its result (50 - 60 us of launch calls, during which the device would stand idle).  The guess is checked below; a
     * wrong one costs the device one search of a short file (the guess is not made for long ones) before the right one. */
    if (a->tab_hit && e->spec_valid && e->table_cache && (uint64_t)a->nsf * C <= SPECULATE_MAX_GROUPS) {
      const uint32_t guess = (e->file_or_word != 0) ? e->file_or_word : e->spec_or;
      if (guess != 0 && lshift_of(guess, bps, &e->lshift) == 0) {
        decide_routes(e, a, guess, 0);
        a->spec = 1;
        RCCHK(launch_searches(e, a, 0, a->trace, a->t_begin));
      }
    }
    HIPCHK(hipEventSynchronize(e->ev_prep));
    if (e->h_or[1] != 0) {
      HIPCHK(hipMemcpyAsync(e->h_nz.ptr, e->d_nz.ptr, (size_t)head_words * 8, hipMemcpyDeviceToHost, e->stream));
      HIPCHK(hipStreamSynchronize(e->stream));
      e->nz_ones_words = 0;
      rebuild = 1;
    } else {
      if (e->nz_ones_cap != e->h_nz.cap) { e->nz_ones_cap = e->h_nz.cap; e->nz_ones_words = 0; }   /* reallocated since (a new block may reuse the address) */
      if (e->nz_ones_words < head_words) {
        memset((uint64_t*)e->h_nz.ptr + e->nz_ones_words, 0xFF, (size_t)(head_words - e->nz_ones_words) * 8);
      }
      e->nz_ones_words = head_words;       /* the words behind it were just overwritten by the tail copy */
    }
  }
  }
  nz = (const uint64_t*)e->h_nz.ptr;
  ((uint64_t*)e->h_nz.ptr)[nwords] = 0; ((uint64_t*)e->h_nz.ptr)[nwords + 1] = 0;
  if (e->file_or_word != 0) {
    if ((e->h_or[0] & ~e->file_or_word) != 0) { return SLA_APIRESULT_INVALID_ARGUMENT; }   /* not a range of that file */
    e->h_or[0] = e->file_or_word;
  }
  PTRACE("prepass + mask on the host");

  RCCHK(lshift_of(e->h_or[0], bps, &e->lshift));

  /* without all-zero mask words only the last super-frame can still be silent (its minimum block length is what
   * is left of the file) */
  if (!rebuild && a->nsf > 0) {
    const sframe_t* f = &a->sf[a->nsf - 1];
    const uint32_t remain = n - f->start;
    const uint32_t window = (maxb < remain) ? maxb : remain;
    const uint32_t min_blk = (SLAI_MIN_BLOCK < remain) ? SLAI_MIN_BLOCK : remain;
    if (slai_zero_run(nz, f->start, window) >= min_blk) { rebuild = 1; }
  }
  if (rebuild) {
    if (!a->borrowed) { free(a->sf); free(a->shapes); }
    a->borrowed = 0; e->tab_valid = 0;                      /* the tables with silence overwrite the kept ones' device copies */
    a->sf = NULL; a->shapes = NULL;
    a->nsf = 0; a->nshapes = 0; a->ncands = 0; a->nsgroups = 0; a->nslots = 0; a->nxg = 0;
    a->blocks_bound = 0; a->lchunks_bound = 0;
    RCCHK(build_tables(e, a, nz));
    RCCHK(pipeline_reserve(e, a));
    RCCHK(upload_search_tables(e, a));
  }
  {
    const int silence = (rebuild || e->h_or[1] != 0);
    if (a->spec && (silence || e->h_or[0] != a->or_word)) {
      /* guessed wrong: the searches go out again.  Whatever the first run queued on the block stream behind its tables (the
       * prelaunched block kernels: they still read the descriptors, counters and candidate slots the second run rewrites) must
       * have drained first: the search stream waits for that stream's tail (the two runs were unordered) */
      if (a->prelaunched) {
        const hipStream_t ps = a->one_stream ? e->stream : e->stream2;
        if (ps != e->stream) {
          HIPCHK(hipEventRecord(e->ev_prep, ps));
          HIPCHK(hipStreamWaitEvent(e->stream, e->ev_prep, 0));
        }
        a->prelaunched = 0;
      }
      a->spec = 0; e->spec_misses++;
    }
    decide_routes(e, a, e->h_or[0], silence);
    e->spec_valid = !silence; e->spec_or = e->h_or[0];
  }
  return 0;
}

/* size every device / pinned buffer of the block and tail stages for the whole file */
static int pipeline_reserve(struct SLAEncoder* e, const actx_t* a)
{
  const uint32_t C = e->wave_format.num_channels, order = e->encode_param.parcor_order, O1 = order + 1, O2 = order + 2;
  const size_t nslots = (size_t)a->blocks_bound * C + 1;
  const uint32_t fft_size = slai_fft_plan_size(e->fft);
  /* search */
  RCCHK(dev_reserve(&e->d_groups, sizeof(sla_hip_lpc_group) * (a->nsgroups + 1)));
  RCCHK(dev_reserve(&e->d_cands, sizeof(sla_hip_lpc_cand) * (a->ncands + 1)));
  RCCHK(dev_reserve(&e->d_lpc_out, sizeof(double) * ((size_t)a->nslots + 1) * O2));
  RCCHK(pin_reserve(&e->h_lpc_out, sizeof(double) * ((size_t)a->nslots + 1) * O2));
  RCCHK(dev_reserve(&e->d_xgroups, sizeof(sla_hip_lpc_group) * ((size_t)a->nxg + 1)));
  {
    const size_t live = (size_t)a->nxg / C + 1;
    RCCHK(dev_reserve(&e->d_parts, sizeof(uint32_t) * live * SLA_HIP_PLAN_NODES));
    RCCHK(dev_reserve(&e->d_nparts, sizeof(uint32_t) * live));
    RCCHK(dev_reserve(&e->d_pstatus, sizeof(uint32_t) * live));
    RCCHK(pin_reserve(&e->h_parts, sizeof(uint32_t) * live * SLA_HIP_PLAN_NODES));
    RCCHK(pin_reserve(&e->h_nparts, sizeof(uint32_t) * live));
    RCCHK(pin_reserve(&e->h_pstatus, sizeof(uint32_t) * live));
  }
  if (e->search_exact && sla_hip_search_exact_lags(order) != 0) {      /* (whether the search may use them is decided once the prepass is in) */
    RCCHK(dev_reserve(&e->d_tile_sums, sizeof(double) * ((size_t)a->nxg + 1) * SLA_HIP_XTILES * 2 * sla_hip_search_exact_lags(order)));
  }
  /* blocks */
  RCCHK(pin_reserve(&e->h_bgroups, sizeof(sla_hip_lpc_group) * nslots));
  RCCHK(pin_reserve(&e->h_bcands, sizeof(sla_hip_lpc_cand) * nslots));
  RCCHK(pin_reserve(&e->h_chunks, sizeof(sla_hip_lattice_chunk) * ((size_t)a->lchunks_bound + 1)));
  RCCHK(pin_reserve(&e->h_acf_jobs, sizeof(sla_hip_acf_job) * nslots));
  RCCHK(pin_reserve(&e->h_jobs, sizeof(sla_hip_tail_job) * nslots));
  RCCHK(pin_reserve(&e->h_blk_out, sizeof(double) * nslots * O2));
  RCCHK(pin_reserve(&e->h_code, sizeof(int32_t) * nslots * O1));
  RCCHK(pin_reserve(&e->h_kint, sizeof(int32_t) * nslots * O1));
  RCCHK(pin_reserve(&e->h_rshift, sizeof(uint32_t) * nslots));
  RCCHK(pin_reserve(&e->h_acf, sizeof(double) * nslots * SLAI_LTM_ACF_HEAD));
  RCCHK(pin_reserve(&e->h_fold, sizeof(uint64_t) * nslots));
  RCCHK(dev_reserve(&e->d_bgroups, sizeof(sla_hip_lpc_group) * nslots));
  RCCHK(dev_reserve(&e->d_bcands, sizeof(sla_hip_lpc_cand) * nslots));
  RCCHK(dev_reserve(&e->d_chunks, sizeof(sla_hip_lattice_chunk) * ((size_t)a->lchunks_bound + 1)));
  RCCHK(dev_reserve(&e->d_acf_jobs, sizeof(sla_hip_acf_job) * nslots));
  RCCHK(dev_reserve(&e->d_jobs, sizeof(sla_hip_tail_job) * nslots));
  RCCHK(dev_reserve(&e->d_blk_out, sizeof(double) * nslots * O2));
  RCCHK(dev_reserve(&e->d_code, sizeof(int32_t) * nslots * O1));
  RCCHK(dev_reserve(&e->d_kint, sizeof(int32_t) * nslots * O1));
  RCCHK(dev_reserve(&e->d_rshift, sizeof(uint32_t) * nslots));
  RCCHK(dev_reserve(&e->d_cert_flag, sizeof(uint32_t) * nslots));
  RCCHK(dev_reserve(&e->d_fb_list, sizeof(uint32_t) * nslots));
  RCCHK(dev_reserve(&e->d_fb_count, sizeof(uint32_t) * MAX_CHUNKS));
  RCCHK(pin_reserve(&e->h_cert_flag, sizeof(uint32_t) * (nslots + MAX_CHUNKS)));
  RCCHK(dev_reserve(&e->d_acf, sizeof(double) * nslots * SLAI_LTM_ACF_HEAD));
  RCCHK(dev_reserve(&e->d_fold, sizeof(uint64_t) * nslots));
  if (sizeof(double) * (size_t)fft_size > SLA_HIP_LDS_BUDGET) {
    RCCHK(dev_reserve(&e->d_acf_scratch, sizeof(double) * (size_t)fft_size * 512));
  }
  if (e->user_res1 != NULL && e->user_stride != e->stride) { return SLA_APIRESULT_INVALID_ARGUMENT; }
  if (e->user_res1 == NULL) {
    RCCHK(dev_reserve(&e->d_res1, sizeof(int32_t) * (size_t)C * e->stride));
    RCCHK(dev_reserve(&e->d_res2, sizeof(int32_t) * (size_t)C * e->stride));
  }
  /* host result arrays */
  if (e->bc_cap < nslots) {
    e->bc_cap = nslots;
    e->bc = (blkch_t*)realloc(e->bc, sizeof(blkch_t) * e->bc_cap);
  }
  if (e->coef_cap < nslots * O1) {
    e->coef_cap = nslots * O1;
    e->parcor = (double*)realloc(e->parcor, sizeof(double) * e->coef_cap);
    e->code = (int32_t*)realloc(e->code, sizeof(int32_t) * e->coef_cap);
    e->kint = (int32_t*)realloc(e->kint, sizeof(int32_t) * e->coef_cap);
    e->parcor_exact = (uint8_t*)realloc(e->parcor_exact, e->coef_cap);
  }
  if (e->bc == NULL || e->parcor == NULL || e->code == NULL || e->kint == NULL || e->parcor_exact == NULL) { return SLA_APIRESULT_NG; }
  memset(e->bc, 0, sizeof(blkch_t) * nslots);
  /* tables that never change while kernels are in flight */
  if (e->win_dirty) {
    /* (new block lengths since the last file: rare.  win_host is ordinary memory that the next file may realloc, so the
     * copy is waited for -- together with whatever else the search stream is doing) */
    RCCHK(dev_reserve(&e->d_winpool, sizeof(double) * (e->win_count + 1)));
    HIPCHK(hipMemcpyAsync(e->d_winpool.ptr, e->win_host, sizeof(double) * e->win_count, hipMemcpyHostToDevice, e->stream));
    HIPCHK(hipStreamSynchronize(e->stream));
    e->win_dirty = 0;
  }
  if (!e->twiddle_ready) {
    double* tw = (double*)malloc(sizeof(double) * SLA_HIP_TWIDDLE_DOUBLES(fft_size));
    if (tw == NULL) { return SLA_APIRESULT_NG; }
    slai_fft_plan_export(e->fft, tw);
    RCCHK(dev_reserve(&e->d_twiddle, sizeof(double) * SLA_HIP_TWIDDLE_DOUBLES(fft_size)));
    HIPCHK(hipMemcpy(e->d_twiddle.ptr, tw, sizeof(double) * SLA_HIP_TWIDDLE_DOUBLES(fft_size), hipMemcpyHostToDevice));
    free(tw);
    e->twiddle_ready = 1;
  }
  return 0;
}

/* stage 1 of chunk c: enqueue the partition-search kernel and the copy of its results */
static int search_launch(struct SLAEncoder* e, actx_t* a, uint32_t c)
{
  const chunk_t* k = &a->ck[c];
  const uint32_t order = e->encode_param.parcor_order, O2 = order + 2;
  const uint32_t ms = (e->encode_param.ch_process_method == SLA_CHPROCESSMETHOD_STEREO_MS);
  hipEvent_t* ev = a->ev + (size_t)c * EV_PER_CHUNK;
  const uint32_t ng = k->grp_hi - k->grp_lo;
  const uint32_t C = e->wave_format.num_channels;
  HIPCHK(hipEventRecord(ev[EV_SEARCH_S], e->stream));
  if (ng > 0) {
    const uint32_t nx = k->xg_hi - k->xg_lo;
    sla_hip_lpc_group* dx = (sla_hip_lpc_group*)e->d_xgroups.ptr + k->xg_lo;
    if (a->exact) {
      /* unit of the samples the search sees: 2^(ntz-31), halved by the mid channel's /2 */
      const int ntz = __builtin_ctz(a->or_word);
      const double limit = ldexp(1.0, e->exact_bits + 2 * (ntz - 31 - (int)ms));
      HIPCHK(hipEventRecord(ev[EV_SEARCH_S], e->stream));
      /* windows over the exactness limit (loud material wider than 16 bits): with the device plan their tile sums are
       * kept and carry a certificate (sla_hip.h), k_plan accepts a partition no admissible rounding could change and
       * flags the rest, which plan_chunk reruns as serial chains.  Without the device plan (or with the certificate
       * switched off) they are flagged at once and take the chains here, without a host round trip. */
      const double cert = e->device_plan ? e->cert_safety : 0.0;
      sla_hip_launch_extra xs;
      memset(&xs, 0, sizeof(xs));
      if (a->clear_in_kernel && a->clear_ptr[0] != NULL) {      /* words this analysis wants zeroed before its kernels use them: once */
        int q;
        for (q = 0; q < 3; q++) { xs.clear_ptr[q] = a->clear_ptr[q]; xs.clear_words[q] = a->clear_words[q]; a->clear_ptr[q] = NULL; }
      }
      RCCHK(sla_hip_launch_search_exact_x(e->pcm_dev, e->stride, ms, order, dx, nx, a->max_window, a->max_xcands, (const sla_hip_lpc_cand*)e->d_cands.ptr,
                                          (double*)e->d_tile_sums.ptr + (size_t)k->xg_lo * SLA_HIP_XTILES * 2 * sla_hip_search_exact_lags(order),
                                          (double*)e->d_lpc_out.ptr, limit, cert, (uint32_t*)e->d_or.ptr + 8 + c, e->stream, &xs));
      if (!(cert > 0.0)) { RCCHK(search_groups_ready(e, a)); }
      if (!(cert > 0.0))
      RCCHK(sla_hip_launch_lpc_rerun(e->pcm_dev, e->stride, ms, order, (const sla_hip_lpc_group*)e->d_groups.ptr + k->grp_lo, ng,
                                     a->max_window, a->max_cpg, (const sla_hip_lpc_cand*)e->d_cands.ptr, (double*)e->d_lpc_out.ptr,
                                     (uint32_t*)e->d_or.ptr + 2, e->stream));
    } else {
      sla_hip_lpc_group* dg = (sla_hip_lpc_group*)e->d_groups.ptr + k->grp_lo;
      RCCHK(search_groups_ready(e, a));
      HIPCHK(hipEventRecord(ev[EV_SEARCH_S], e->stream));
      RCCHK(sla_hip_launch_lpc(e->pcm_dev, e->stride, ms, order, dg, ng, a->max_window, a->max_cpg,
                               (const sla_hip_lpc_cand*)e->d_cands.ptr, NULL, (double*)e->d_lpc_out.ptr, NULL, NULL, NULL, e->stream));
    }
    HIPCHK(hipEventRecord(ev[EV_SEARCH_E], e->stream));
    if (e->device_plan) {
      /* plan on the device; only the block lengths come back (the candidates' doubles follow on demand) */
      const uint32_t live_lo = k->xg_lo / C, live = nx / C;
      RCCHK(sla_hip_launch_plan(dx, live, C, order, e->wave_format.bit_per_sample, (const sla_hip_lpc_cand*)e->d_cands.ptr,
                                (double*)e->d_lpc_out.ptr, (uint32_t*)e->d_parts.ptr + (size_t)live_lo * SLA_HIP_PLAN_NODES,
                                (uint32_t*)e->d_nparts.ptr + live_lo, (uint32_t*)e->d_pstatus.ptr + live_lo, e->stream));
      HIPCHK(hipEventRecord(ev[EV_PLANNED], e->stream));
      if (a->expand) {
        /* the chunk's block table, written on the device right behind the plan; two counts (polled in page-locked memory)
         * are all the host needs to launch the block stage -- its own copy of the tables follows under those kernels */
        volatile uint32_t* hc = (volatile uint32_t*)e->h_counts.ptr + 4 * (size_t)c;
        const uint32_t bps = e->wave_format.bit_per_sample;
        hc[0] = hc[1] = hc[2] = hc[3] = 0;
        if (c == 0 && !a->clear_in_kernel) { HIPCHK(hipMemsetAsync(e->d_run.ptr, 0, 16, e->stream)); }      /* the running block / group numbers restart */
        RCCHK(sla_hip_launch_expand_masked((const sla_hip_superframe*)e->d_sframes.ptr + k->sf_lo, k->sf_hi - k->sf_lo,
                                    (const uint32_t*)e->d_parts.ptr, (const uint32_t*)e->d_nparts.ptr, (const uint32_t*)e->d_pstatus.ptr,
                                    C, 32 - bps + e->lshift, (const uint32_t*)e->d_winmap.ptr, (const uint32_t*)e->d_winmap.ptr + e->winmap_entries,
                                    e->winmap_entries, (uint32_t*)e->d_run.ptr, (uint32_t*)e->d_expref.ptr, (sla_hip_lpc_group*)e->d_bgroups.ptr, (sla_hip_lpc_cand*)e->d_bcands.ptr,
                                    (sla_hip_acf_job*)e->d_acf_jobs.ptr, a->blocks_bound * C, (uint32_t*)e->h_counts.ptr + 4 * (size_t)c,
                                    e->expand_seq, a->silence ? (const uint64_t*)e->d_nz.ptr : NULL, e->stream));
        HIPCHK(hipEventRecord(ev[EV_EXPANDED], e->stream));
        /* Short files: the first two kernels of the block stage go out right here, sized for the most groups the chunk can
         * have and reading the number from the device (sla_hip_launch_extra.d_group_count) -- the device starts on them the moment the
         * tables are written instead of 40 - 50 us later, when the host has seen the counts and made its first launches;
         * by then it only has to queue the lattice behind them.  (Long files: their idle workgroups would cost more.) */
        a->prelaunched = 0;
        if (c == 0 && a->nchunks == 1 && e->cert_now && !(e->fuse_lattice && order <= 64) && (uint64_t)a->nsf * C <= SPECULATE_MAX_GROUPS
            && e->device_ltm && e->prelaunch) {
          const uint32_t ms2 = (e->encode_param.ch_process_method == SLA_CHPROCESSMETHOD_STEREO_MS);
          const hipStream_t ps = a->one_stream ? e->stream : e->stream2;      /* the stream blocks_launch will continue on */
          HIPCHK(hipStreamWaitEvent(ps, ev[EV_EXPANDED], 0));
          HIPCHK(hipEventRecord(ev[EV_LPCB_S], ps));
          sla_hip_launch_extra xt;
          memset(&xt, 0, sizeof(xt));
          xt.d_span = SPAN_SLOT(e, c, 0); xt.d_group_count = (const uint32_t*)e->d_run.ptr;
          RCCHK(sla_hip_launch_lpc_blocks_cert_x(e->pcm_dev, e->stride, ms2, order, (const sla_hip_lpc_group*)e->d_bgroups.ptr, a->blocks_bound * C,
                                                 a->max_window, (const sla_hip_lpc_cand*)e->d_bcands.ptr, (const double*)e->d_winpool.ptr,
                                                 (double*)e->d_blk_out.ptr, (int32_t*)e->d_code.ptr, (int32_t*)e->d_kint.ptr,
                                                 (uint32_t*)e->d_rshift.ptr, (uint32_t*)e->d_cert_flag.ptr,
                                                 (uint32_t*)e->d_fb_list.ptr, (uint32_t*)e->d_fb_count.ptr + c,
                                                 e->block_cert_safety, bps, ps, &xt));
          HIPCHK(hipEventRecord(ev[EV_LPCB_E], ps));
          a->prelaunched = 1;
        }
      }
      {
        /* The three small result copies stay on the search stream, in order behind k_plan.  On the download stream
         * they would sit there waiting for every chunk's search in turn, in front of whatever shares that stream's
         * hardware queue (measured with one queue per stream: 2.9 instead of 3.05 ms per C2 step;
         * SLA_HIP_PLAN_COPY=down restores the download stream). */
        /* (one chunk with everything on the search stream: there the copies would sit in front of the block stage) */
        const hipStream_t st = (e->plan_copy_down || a->one_stream) ? e->stream_down : e->stream;
        if (st != e->stream) { HIPCHK(hipStreamWaitEvent(st, ev[EV_PLANNED], 0)); }
        HIPCHK(hipMemcpyAsync((uint32_t*)e->h_parts.ptr + (size_t)live_lo * SLA_HIP_PLAN_NODES, (uint32_t*)e->d_parts.ptr + (size_t)live_lo * SLA_HIP_PLAN_NODES,
                              sizeof(uint32_t) * (size_t)live * SLA_HIP_PLAN_NODES, hipMemcpyDeviceToHost, st));
        HIPCHK(hipMemcpyAsync((uint32_t*)e->h_nparts.ptr + live_lo, (uint32_t*)e->d_nparts.ptr + live_lo, sizeof(uint32_t) * live, hipMemcpyDeviceToHost, st));
        HIPCHK(hipMemcpyAsync((uint32_t*)e->h_pstatus.ptr + live_lo, (uint32_t*)e->d_pstatus.ptr + live_lo, sizeof(uint32_t) * live, hipMemcpyDeviceToHost, st));
        HIPCHK(hipEventRecord(ev[EV_SEARCH_DONE], st));      /* what the host waits for */
      }
      return 0;
    } else {
      HIPCHK(hipMemcpyAsync((double*)e->h_lpc_out.ptr + (size_t)k->slot_lo * O2, (double*)e->d_lpc_out.ptr + (size_t)k->slot_lo * O2,
                            sizeof(double) * (size_t)(k->slot_hi - k->slot_lo) * O2, hipMemcpyDeviceToHost, e->stream));
    }
  } else {
    HIPCHK(hipEventRecord(ev[EV_SEARCH_E], e->stream));
  }
  HIPCHK(hipEventRecord(ev[EV_SEARCH_DONE], e->stream));
  return 0;
}

/* host: plan of chunk c -> block table entries [blk_lo, blk_hi) */
static int plan_chunk(struct SLAEncoder* e, actx_t* a, uint32_t c)
{
  chunk_t* k = &a->ck[c];
  const uint64_t* nz = e->mask_absent ? NULL : (const uint64_t*)e->h_nz.ptr;      /* NULL: nothing is silent (slai_zero_run) */
  plan_ctx_t ctx;
  uint32_t i;
  int certified_only = 0;
  ctx.e = e; ctx.a = a; ctx.sf_lo = k->sf_lo;
  ctx.cands = (const sla_hip_lpc_cand*)e->h_cands.ptr; ctx.out = (const double*)e->h_lpc_out.ptr;
  if (e->device_plan) {
    /* super-frames the device could not certify (or whose search was flagged) need their candidates' doubles */
    const uint32_t C = e->wave_format.num_channels, O2 = e->encode_param.parcor_order + 2;
    const uint32_t* st = (const uint32_t*)e->h_pstatus.ptr;
    uint32_t open_frames = 0, chain_frames = 0;
    for (i = k->sf_lo; i < k->sf_hi; i++) {
      if (a->sf[i].shape != 0xFFFFFFFFu && st[a->sf[i].xg / C] != 0) { open_frames++; if (st[a->sf[i].xg / C] == 2) { chain_frames++; } }
    }
    e->host_planned += open_frames;
    certified_only = (open_frames == 0);
    if (chain_frames != 0) {
      /* certified tile sums that did not certify: k_plan flagged their candidates, the chain kernel redoes exactly those
       * groups in the reference's order (the others return at once) before the doubles come home */
      const uint32_t order = e->encode_param.parcor_order;
      const uint32_t ms = (e->encode_param.ch_process_method == SLA_CHPROCESSMETHOD_STEREO_MS);
      RCCHK(search_groups_ready(e, a));
      RCCHK(sla_hip_launch_lpc_rerun(e->pcm_dev, e->stride, ms, order, (const sla_hip_lpc_group*)e->d_groups.ptr + k->grp_lo, k->grp_hi - k->grp_lo,
                                     a->max_window, a->max_cpg, (const sla_hip_lpc_cand*)e->d_cands.ptr, (double*)e->d_lpc_out.ptr,
                                     (uint32_t*)e->d_or.ptr + 2, e->stream));
    }
    if (open_frames != 0) {
      HIPCHK(hipMemcpyAsync((double*)e->h_lpc_out.ptr + (size_t)k->slot_lo * O2, (double*)e->d_lpc_out.ptr + (size_t)k->slot_lo * O2,
                            sizeof(double) * (size_t)(k->slot_hi - k->slot_lo) * O2, hipMemcpyDeviceToHost, e->stream));
      HIPCHK(hipStreamSynchronize(e->stream));
    }
  }
  if (e->device_plan && certified_only) {
    for (i = 0; i < k->sf_hi - k->sf_lo; i++) { plan_one(&ctx, i); }        /* a copy per super-frame: not worth waking the pool */
  } else {
    parallel_for(e->pool, k->sf_hi - k->sf_lo, plan_one, &ctx);
  }
  k->blk_lo = e->num_blocks;
  for (i = k->sf_lo; i < k->sf_hi; i++) {
    const sframe_t* f = &a->sf[i];
    if (f->shape == 0xFFFFFFFFu) {
      RCCHK(blocks_push(e, f->start, f->window, SLAI_BLK_SILENT));
    } else {
      uint32_t p, at = f->start;
      if (a->status[i] != 0) { return a->status[i]; }
      for (p = 0; p < a->nparts[i]; p++) {
        const uint32_t len = a->parts[(size_t)i * SLAI_MAX_NODES + p];
        RCCHK(blocks_push(e, at, len, slai_range_is_zero(nz, at, len) ? SLAI_BLK_SILENT : SLAI_BLK_COMPRESS));
        at += len;
      }
    }
  }
  k->blk_hi = e->num_blocks;
  if (k->blk_hi > a->blocks_bound) { return SLA_APIRESULT_NG; }
  return 0;
}


/* device long-term mode: k_tail over the jobs [lo, hi) (job == block group) right behind k_ltm_solve on the kernel
 * stream; folded sums and the solved pitch/taps go home on the download stream */
static int tail_enqueue(struct SLAEncoder* e, actx_t* a, uint32_t c, uint32_t lo, uint32_t hi)
{
  const uint32_t ntaps = e->encode_param.longterm_order, lms = e->encode_param.lms_order_per_filter;
  hipEvent_t* ev = a->ev + (size_t)c * EV_PER_CHUNK;
  /* one tail per chunk: on its own stream, so that the serial LMS chains (few waves, latency-bound) run beside the next
   * chunk's throughput-bound kernels; one tail for the file: behind the last solve on the kernel stream */
  hipStream_t ts = (a->one_stream && a->launched[0]) ? e->stream : (e->single_tail ? e->stream2 : e->stream3);
  a->tail_stream = ts;
  if (!e->single_tail) {
    HIPCHK(hipEventRecord(ev[EV_SOLVED], e->stream2));
    HIPCHK(hipStreamWaitEvent(ts, ev[EV_SOLVED], 0));
  } else if (e->alt_now) {
    uint32_t cc;
    for (cc = 1; cc < a->nchunks; cc += 2) {
      if (a->ck[cc].bg_hi > a->ck[cc].bg_lo) { HIPCHK(hipStreamWaitEvent(ts, a->ev[(size_t)cc * EV_PER_CHUNK + EV_SOLVED], 0)); }
    }
  }
  HIPCHK(hipEventRecord(ev[EV_TAIL_S], ts));
  if (hi > lo) {
    sla_hip_tail_job* dj = (sla_hip_tail_job*)e->d_jobs.ptr + lo;
    sla_hip_launch_extra xt;
    memset(&xt, 0, sizeof(xt));
    xt.d_span = SPAN_SLOT(e, c, 3);
    RCCHK(sla_hip_launch_tail_x(RES1(e), RES2(e), e->stride, dj, hi - lo, ntaps, lms, (uint64_t*)e->d_fold.ptr + lo, ts, &xt));
  }
  HIPCHK(hipEventRecord(ev[EV_TAIL_E], ts));
  if (hi > lo) {
    HIPCHK(hipStreamWaitEvent(e->stream_down, ev[EV_TAIL_E], 0));
    HIPCHK(hipMemcpyAsync((uint64_t*)e->h_fold.ptr + lo, (uint64_t*)e->d_fold.ptr + lo, sizeof(uint64_t) * (hi - lo), hipMemcpyDeviceToHost, e->stream_down));
    HIPCHK(hipMemcpyAsync((sla_hip_tail_job*)e->h_jobs.ptr + lo, (sla_hip_tail_job*)e->d_jobs.ptr + lo, sizeof(sla_hip_tail_job) * (hi - lo),
                          hipMemcpyDeviceToHost, e->stream_down));
  }
  HIPCHK(hipEventRecord(ev[EV_TAIL_DONE], e->stream_down));
  return 0;
}

/* stage 2 of chunk c: windowed LPC + quantiser, lattice, long-term FFT for blocks [blk_lo, blk_hi).
 * mode 0: host tables, their upload, the launches.  With device-written tables (k_expand) the two halves come apart:
 * mode 1 launches from the counts in a->dev_lo[c] alone, mode 2 -- later, under those kernels -- builds the host's
 * own tables from the partitions that have come home meanwhile and checks that both sides numbered alike. */
static int blocks_launch(struct SLAEncoder* e, actx_t* a, uint32_t c, int mode)
{
  chunk_t* k = &a->ck[c];
  const uint32_t C = e->wave_format.num_channels, bps = e->wave_format.bit_per_sample;
  const uint32_t order = e->encode_param.parcor_order, O1 = order + 1, O2 = order + 2;
  const uint32_t ms = (e->encode_param.ch_process_method == SLA_CHPROCESSMETHOD_STEREO_MS);
  const uint32_t shift = 32 - bps + e->lshift;
  extern uint32_t sla_hip_lattice_chunk_samples(uint32_t order);
  const uint32_t chunk_samples = sla_hip_lattice_chunk_samples(order);
  const uint32_t fft_size = slai_fft_plan_size(e->fft);
  sla_hip_lpc_group* groups = (sla_hip_lpc_group*)e->h_bgroups.ptr;
  sla_hip_lpc_cand* cands = (sla_hip_lpc_cand*)e->h_bcands.ptr;
  sla_hip_lattice_chunk* lch = (sla_hip_lattice_chunk*)e->h_chunks.ptr;
  sla_hip_acf_job* acf_jobs = (sla_hip_acf_job*)e->h_acf_jobs.ptr;
  hipEvent_t* ev = a->ev + (size_t)c * EV_PER_CHUNK;
  uint32_t b, ch, max_window = 1, ng, nl;
  const int fused = (e->fuse_lattice && order <= 64);
  const int use_cert = e->cert_now;
  const int pre = (use_cert && mode == 1 && c == 0 && a->prelaunched && !(e->fuse_lattice && order <= 64));
  const size_t cnt_base = (size_t)a->blocks_bound * C + 1;      /* h_cert_flag: per-slot flags, then one fallback count per chunk */
  /* option alt_streams: odd chunks run their block stage on the third stream, beside the even chunks' (not when the FFT
   * works in the shared global scratch) */
  hipStream_t bs = (mode == 1 && a->one_stream) ? e->stream
                   : (e->alt_now && e->device_ltm && e->single_tail && (c & 1u)
                      && sizeof(double) * (size_t)fft_size <= SLA_HIP_LDS_BUDGET) ? e->stream3 : e->stream2;
  size_t slot_lo, nsl;

  if (mode == 1) {
    k->blk_lo = a->dev_lo[c][0]; k->blk_hi = a->dev_lo[c][1]; k->bg_lo = a->dev_lo[c][2]; k->bg_hi = a->dev_lo[c][3];
    k->lc_lo = k->lc_hi = a->nlc;
    max_window = a->max_window;
    if (e->device_ltm) { k->job_lo = k->bg_lo; k->job_hi = k->bg_hi; }
  }
  /* pass 1: what k_lpc_blocks needs (one group per block and channel).  The descriptors of the lattice and FFT
   * launches are built in pass 2, while k_lpc_blocks is already running. */
  if (mode != 1) {
  k->bg_lo = a->nbg; k->lc_lo = a->nlc;
  for (b = k->blk_lo; b < k->blk_hi; b++) {
    const blk_t* blk = &e->blk[b];
    uint32_t woff = 0;
    if (blk->type == SLAI_BLK_SILENT) { continue; }
    if (blk->nsmpl > MAX_ANALYSIS_WINDOW) { return SLA_APIRESULT_EXCEED_HANDLE_CAPACITY; }
    RCCHK(window_offset(e, blk->nsmpl, &woff));
    if (e->win_dirty) { return SLA_APIRESULT_NG; }      /* every length was tabulated in pipeline_prepare */
    for (ch = 0; ch < C; ch++) {
      const uint32_t g = a->nbg++;
      sla_hip_lpc_group* gr = &groups[g];
      gr->pcm_off = blk->start; gr->num_samples = blk->nsmpl; gr->channel = ch; gr->win_off = woff; gr->int_shift = shift;
      gr->cand_first = g; gr->cand_count = 1; gr->slot_first = b * C + ch; gr->pad_ = 0;
      cands[g].start = 0; cands[g].len = blk->nsmpl;
      a->grp_of_slot[(size_t)b * C + ch] = g;
      if (e->device_ltm) { a->job_blk[g] = b; a->job_ch[g] = ch; a->job_grp[g] = g; }      /* tail job g == group g */
    }
    if (blk->nsmpl > max_window) { max_window = blk->nsmpl; }
  }
  k->bg_hi = a->nbg;
  if (e->device_ltm) { k->job_lo = k->bg_lo; k->job_hi = k->bg_hi; a->njobs = a->nbg; }
  }
  if (mode == 2) {
    k->lc_hi = a->nlc;
    return (k->blk_lo == a->dev_lo[c][0] && k->blk_hi == a->dev_lo[c][1] && k->bg_lo == a->dev_lo[c][2] && k->bg_hi == a->dev_lo[c][3])
           ? 0 : SLA_APIRESULT_NG;
  }
  ng = k->bg_hi - k->bg_lo;
  slot_lo = (size_t)k->blk_lo * C; nsl = (size_t)(k->blk_hi - k->blk_lo) * C;

  if (ng == 0) { k->lc_hi = a->nlc; HIPCHK(hipEventRecord(ev[EV_LPCB_S], bs)); }
  if (ng > 0) {
    sla_hip_lpc_group* dg = (sla_hip_lpc_group*)e->d_bgroups.ptr + k->bg_lo;
    sla_hip_lattice_chunk* dl = (sla_hip_lattice_chunk*)e->d_chunks.ptr + k->lc_lo;
    sla_hip_acf_job* da = (sla_hip_acf_job*)e->d_acf_jobs.ptr + k->bg_lo;
    uint32_t slots = 0;
    if (mode == 0) {
    HIPCHK(hipMemcpyAsync(dg, groups + k->bg_lo, sizeof(sla_hip_lpc_group) * ng, hipMemcpyHostToDevice, e->stream_up));
    HIPCHK(hipMemcpyAsync((sla_hip_lpc_cand*)e->d_bcands.ptr + k->bg_lo, cands + k->bg_lo, sizeof(sla_hip_lpc_cand) * ng, hipMemcpyHostToDevice, e->stream_up));
    /* (k_lpc_blocks writes every output slot of its groups; slots of silent blocks are never read) */
    /* uploads and result downloads travel on their own streams: the kernel stream runs kernel after kernel */
    HIPCHK(hipEventRecord(ev[EV_UPLOADED], e->stream_up));
    HIPCHK(hipStreamWaitEvent(bs, ev[EV_UPLOADED], 0));
    } else {
      HIPCHK(hipStreamWaitEvent(bs, ev[EV_EXPANDED], 0));      /* the tables were written on the search stream */
    }
    sla_hip_launch_extra xb;                     /* the block stage's launch records its execution span in slot 0 */
    memset(&xb, 0, sizeof(xb));
    xb.d_span = SPAN_SLOT(e, c, 0);
    if (!pre) { HIPCHK(hipEventRecord(ev[EV_LPCB_S], bs)); }
    if (fused) {
      RCCHK(sla_hip_launch_lpc_blocks_x(e->pcm_dev, e->stride, ms, order, dg, ng, max_window,
                                        (const sla_hip_lpc_cand*)e->d_bcands.ptr, (const double*)e->d_winpool.ptr,
                                        (double*)e->d_blk_out.ptr, (int32_t*)e->d_code.ptr, (int32_t*)e->d_kint.ptr,
                                        (uint32_t*)e->d_rshift.ptr, RES1(e), bs, &xb));
    } else if (pre) {
      /* queued with the searches (search_launch), events and span slot included */
    } else if (use_cert) {
      /* any-order autocorrelation + certified quantiser; what does not certify goes through the exact kernels behind it */
      RCCHK(sla_hip_launch_lpc_blocks_cert_x(e->pcm_dev, e->stride, ms, order, dg, ng, max_window,
                                             (const sla_hip_lpc_cand*)e->d_bcands.ptr, (const double*)e->d_winpool.ptr,
                                             (double*)e->d_blk_out.ptr, (int32_t*)e->d_code.ptr, (int32_t*)e->d_kint.ptr,
                                             (uint32_t*)e->d_rshift.ptr, (uint32_t*)e->d_cert_flag.ptr,
                                             (uint32_t*)e->d_fb_list.ptr + k->bg_lo, (uint32_t*)e->d_fb_count.ptr + c,
                                             e->block_cert_safety, bps, bs, &xb));
    } else {
      RCCHK(sla_hip_launch_lpc_x(e->pcm_dev, e->stride, ms, order, dg, ng, max_window, 1,
                                 (const sla_hip_lpc_cand*)e->d_bcands.ptr, (const double*)e->d_winpool.ptr,
                                 (double*)e->d_blk_out.ptr, (int32_t*)e->d_code.ptr, (int32_t*)e->d_kint.ptr,
                                 (uint32_t*)e->d_rshift.ptr, bs, &xb));
    }
    if (!pre) { HIPCHK(hipEventRecord(ev[EV_LPCB_E], bs)); }
    /* pass 2 (k_lpc_blocks is running): lattice chunks and FFT jobs */
    if (mode == 0) {
      uint32_t g = k->bg_lo;
      for (b = k->blk_lo; b < k->blk_hi; b++) {
        const blk_t* blk = &e->blk[b];
        uint32_t at;
        if (blk->type == SLAI_BLK_SILENT) { continue; }
        for (ch = 0; ch < C; ch++, g++) {
          acf_jobs[g].blk_off = blk->start; acf_jobs[g].blk_len = blk->nsmpl; acf_jobs[g].channel = ch;
          (void)at; (void)lch; (void)chunk_samples;      /* the lattice waves are derived from the block groups on the device */
        }
      }
      k->lc_hi = a->nlc;
      nl = k->lc_hi - k->lc_lo;
      if (nl > 0) { HIPCHK(hipMemcpyAsync(dl, lch + k->lc_lo, sizeof(sla_hip_lattice_chunk) * nl, hipMemcpyHostToDevice, e->stream_up)); }
      HIPCHK(hipMemcpyAsync(da, acf_jobs + k->bg_lo, sizeof(sla_hip_acf_job) * ng, hipMemcpyHostToDevice, e->stream_up));
      HIPCHK(hipEventRecord(ev[EV_UPLOADED2], e->stream_up));
      HIPCHK(hipStreamWaitEvent(bs, ev[EV_UPLOADED2], 0));
    }
    if (!fused) {
      xb.d_span = SPAN_SLOT(e, c, 1);
      RCCHK(sla_hip_launch_lattice_groups_x(e->pcm_dev, e->stride, ms, order, dg, ng, max_window, (const int32_t*)e->d_kint.ptr, RES1(e), bs, &xb));
    }
    HIPCHK(hipEventRecord(ev[EV_LAT_E], bs));
    if (sizeof(double) * (size_t)fft_size > SLA_HIP_LDS_BUDGET) { slots = (ng < 512) ? ng : 512; }
    HIPCHK(hipEventRecord(ev[EV_ACF_S], bs));
    xb.d_span = SPAN_SLOT(e, c, 2);
    RCCHK(sla_hip_launch_ltm_acf_x(RES1(e), e->stride, da, ng, fft_size, (const double*)e->d_twiddle.ptr,
                                   (double*)e->d_acf_scratch.ptr, slots,
                                   (double*)e->d_acf.ptr + (size_t)k->bg_lo * SLAI_LTM_ACF_HEAD, SLAI_LTM_ACF_HEAD, bs, &xb));
    HIPCHK(hipEventRecord(ev[EV_ACF_E], bs));
    if (e->device_ltm) {
      /* pitch + taps into the job table k_tail reads; the tail follows on the same stream, no host in between */
      RCCHK(sla_hip_launch_ltm_solve((const double*)e->d_acf.ptr + (size_t)k->bg_lo * SLAI_LTM_ACF_HEAD, dg, ng,
                                     e->encode_param.longterm_order, (sla_hip_tail_job*)e->d_jobs.ptr + k->bg_lo, bs));
      if (!e->single_tail) { RCCHK(tail_enqueue(e, a, c, k->bg_lo, k->bg_hi)); }
      else { HIPCHK(hipEventRecord(ev[EV_SOLVED], bs)); }
    } else {
      HIPCHK(hipStreamWaitEvent(e->stream_down, ev[EV_ACF_E], 0));
      HIPCHK(hipMemcpyAsync((double*)e->h_acf.ptr + (size_t)k->bg_lo * SLAI_LTM_ACF_HEAD, (double*)e->d_acf.ptr + (size_t)k->bg_lo * SLAI_LTM_ACF_HEAD,
                            sizeof(double) * (size_t)ng * SLAI_LTM_ACF_HEAD, hipMemcpyDeviceToHost, e->stream_down));
    }
    /* The LPC results go home while lattice and FFT run: the host decides RAW blocks in the meantime.  Queued behind every
     * kernel launch of the stage: seven more API calls between the block kernels and the lattice launch left the device
     * waiting for the host whenever one of them was slow (seen as 0.3 - 0.6 ms in the lattice stage of some C2 steps). */
    HIPCHK(hipStreamWaitEvent(e->stream_down, ev[EV_LPCB_E], 0));
    HIPCHK(hipMemcpyAsync((double*)e->h_blk_out.ptr + slot_lo * O2, (double*)e->d_blk_out.ptr + slot_lo * O2, sizeof(double) * nsl * O2, hipMemcpyDeviceToHost, e->stream_down));
    HIPCHK(hipMemcpyAsync((int32_t*)e->h_code.ptr + slot_lo * O1, (int32_t*)e->d_code.ptr + slot_lo * O1, sizeof(int32_t) * nsl * O1, hipMemcpyDeviceToHost, e->stream_down));
    HIPCHK(hipMemcpyAsync((int32_t*)e->h_kint.ptr + slot_lo * O1, (int32_t*)e->d_kint.ptr + slot_lo * O1, sizeof(int32_t) * nsl * O1, hipMemcpyDeviceToHost, e->stream_down));
    HIPCHK(hipMemcpyAsync((uint32_t*)e->h_rshift.ptr + slot_lo, (uint32_t*)e->d_rshift.ptr + slot_lo, sizeof(uint32_t) * nsl, hipMemcpyDeviceToHost, e->stream_down));
    if (use_cert) {
      HIPCHK(hipMemcpyAsync((uint32_t*)e->h_cert_flag.ptr + slot_lo, (uint32_t*)e->d_cert_flag.ptr + slot_lo, sizeof(uint32_t) * nsl, hipMemcpyDeviceToHost, e->stream_down));
      HIPCHK(hipMemcpyAsync((uint32_t*)e->h_cert_flag.ptr + cnt_base + c, (uint32_t*)e->d_fb_count.ptr + c, sizeof(uint32_t), hipMemcpyDeviceToHost, e->stream_down));
    }
    HIPCHK(hipEventRecord(ev[EV_LPC_DOWN], e->stream_down));
  } else {
    HIPCHK(hipEventRecord(ev[EV_LPCB_E], bs));
    HIPCHK(hipEventRecord(ev[EV_LPC_DOWN], e->stream_down));
    HIPCHK(hipEventRecord(ev[EV_LAT_E], bs));
    HIPCHK(hipEventRecord(ev[EV_ACF_S], bs));
    HIPCHK(hipEventRecord(ev[EV_ACF_E], bs));
    if (e->device_ltm && !e->single_tail) { RCCHK(tail_enqueue(e, a, c, k->bg_lo, k->bg_lo)); }
  }
  HIPCHK(hipStreamWaitEvent(e->stream_down, ev[EV_ACF_E], 0));
  HIPCHK(hipEventRecord(ev[EV_BLOCK_DONE], e->stream_down));
  return 0;
}

typedef struct { struct SLAEncoder* e; uint32_t blk_lo; } raw_ctx_t;

/* copy one block's device results into the encoder and decide COMPRESS vs RAW */
static void raw_one(void* vctx, uint32_t rel)
{
  raw_ctx_t* c = (raw_ctx_t*)vctx;
  struct SLAEncoder* e = c->e;
  const uint32_t b = c->blk_lo + rel;
  const uint32_t C = e->wave_format.num_channels, bps = e->wave_format.bit_per_sample;
  const uint32_t order = e->encode_param.parcor_order, O1 = order + 1, O2 = order + 2;
  blk_t* blk = &e->blk[b];
  uint32_t ch;
  if (blk->type == SLAI_BLK_SILENT) { return; }
  for (ch = 0; ch < C; ch++) {
    const size_t slot = (size_t)b * C + ch;
    const double* o = (const double*)e->h_blk_out.ptr + slot * O2;
    double est;
    memcpy(e->parcor + slot * O1, o + 1, sizeof(double) * O1);
    memcpy(e->code + slot * O1, (const int32_t*)e->h_code.ptr + slot * O1, sizeof(int32_t) * O1);
    memcpy(e->kint + slot * O1, (const int32_t*)e->h_kint.ptr + slot * O1, sizeof(int32_t) * O1);
    e->bc[slot].rshift = ((const uint32_t*)e->h_rshift.ptr)[slot];
    /* certified route: 0 = certified doubles, 2 = redone by the exact kernels (1 would be a block the fallback missed) */
    {
      /* 4 / 5: an audited pair (option cert_audit): the exact kernels ran on it too -- their doubles are the ones stored -- and
       * found the certified codes equal / different; 3 would be an audited pair the exact kernels never reached */
      const uint32_t flag = e->cert_now ? ((const uint32_t*)e->h_cert_flag.ptr)[slot] : 2u;
      e->parcor_exact[slot] = (uint8_t)(flag == 2u || flag == 4u);
      if (flag == 1u || flag == 3u || (flag & 7u) == 3u || flag == 5u) { e->cert_broken = 1; }
      if (flag == 4u) { __sync_fetch_and_add(&e->audit_ok, 1u); }
      if (flag == 5u) { __sync_fetch_and_add(&e->audit_bad, 1u); }
    }
    est = slai_code_length(o[0], blk->nsmpl, bps, o + 1, order);
    est = (8 * est) / bps;
    if (est >= SLAI_RAW_THRESHOLD) { blk->type = SLAI_BLK_RAW; break; }
  }
}

/* host part 1 of stage 3 (needs only the LPC results): RAW decision per block, list of tail jobs */
static int raw_phase(struct SLAEncoder* e, actx_t* a, uint32_t c)
{
  chunk_t* k = &a->ck[c];
  const uint32_t C = e->wave_format.num_channels;
  uint32_t b, ch;
  if (!e->device_ltm) { k->job_lo = a->njobs; }
  /* RAW decision per block (any channel's estimate >= 0.95, src/SLAEncoder.c:553-565) on the host threads */
  {
    raw_ctx_t rc;
    rc.e = e; rc.blk_lo = k->blk_lo;
    g_trace_t0 = now_ms();
    parallel_for(e->pool, k->blk_hi - k->blk_lo, raw_one, &rc);
    PTRACE("tail: RAW decision");
  }
  if (e->device_ltm) { k->job_lo = k->bg_lo; k->job_hi = k->bg_hi; return 0; }      /* the tail ran (or runs) over every group */
  for (b = k->blk_lo; b < k->blk_hi; b++) {
    if (e->blk[b].type != SLAI_BLK_COMPRESS) { continue; }
    for (ch = 0; ch < C; ch++) {
      a->job_blk[a->njobs] = b; a->job_ch[a->njobs] = ch; a->job_grp[a->njobs] = a->grp_of_slot[(size_t)b * C + ch];
      a->njobs++;
    }
  }
  k->job_hi = a->njobs;
  return 0;
}

/* host part 2 + stage 3 of chunk c: long-term solve from the FFT records, then k_tail */
/* long-term pitch + taps of the jobs [a->ltm_done, upto) from their downloaded autocorrelations (host pool) */
static void ltm_solve(struct SLAEncoder* e, actx_t* a, uint32_t upto)
{
  ltm_ctx_t lc;
  if (upto <= a->ltm_done) { return; }
  lc.e = e; lc.acf = (const double*)e->h_acf.ptr; lc.job_blk = a->job_blk; lc.job_ch = a->job_ch; lc.job_grp = a->job_grp;
  lc.job_lo = a->ltm_done;
  parallel_for(e->pool, upto - a->ltm_done, ltm_one, &lc);
  a->ltm_done = upto;
}

static int tail_launch(struct SLAEncoder* e, actx_t* a, uint32_t c)
{
  chunk_t* k = &a->ck[c];
  const uint32_t C = e->wave_format.num_channels;
  const uint32_t ntaps = e->encode_param.longterm_order, lms = e->encode_param.lms_order_per_filter;
  sla_hip_tail_job* jobs = (sla_hip_tail_job*)e->h_jobs.ptr;
  hipEvent_t* ev = a->ev + (size_t)c * EV_PER_CHUNK;
  uint32_t j, t, nj;
  nj = k->job_hi - k->job_lo;
  ltm_solve(e, a, k->job_hi);
  PTRACE("tail: long-term solve");
  HIPCHK(hipEventRecord(ev[EV_TAIL_S], e->stream3));
  if (nj > 0) {
    sla_hip_tail_job* dj = (sla_hip_tail_job*)e->d_jobs.ptr + k->job_lo;
    for (j = k->job_lo; j < k->job_hi; j++) {
      const blk_t* blk = &e->blk[a->job_blk[j]];
      const blkch_t* bc = &e->bc[(size_t)a->job_blk[j] * C + a->job_ch[j]];
      jobs[j].blk_off = blk->start; jobs[j].blk_len = blk->nsmpl; jobs[j].channel = a->job_ch[j]; jobs[j].pitch = bc->pitch;
      for (t = 0; t < SLAI_MAX_TAPS; t++) { jobs[j].ltm_coef[t] = bc->ltm_q[t]; }
      jobs[j].pad_[0] = jobs[j].pad_[1] = 0;
    }
    PTRACE("tail: jobs built");
    HIPCHK(hipMemcpyAsync(dj, jobs + k->job_lo, sizeof(sla_hip_tail_job) * nj, hipMemcpyHostToDevice, e->stream3));
    HIPCHK(hipEventRecord(ev[EV_TAIL_S], e->stream3));
    {
      sla_hip_launch_extra xt;
      memset(&xt, 0, sizeof(xt));
      xt.d_span = SPAN_SLOT(e, c, 3);
      RCCHK(sla_hip_launch_tail_x(RES1(e), RES2(e), e->stride, dj, nj, ntaps, lms, (uint64_t*)e->d_fold.ptr + k->job_lo, e->stream3, &xt));
    }
    HIPCHK(hipEventRecord(ev[EV_TAIL_E], e->stream3));
    HIPCHK(hipMemcpyAsync((uint64_t*)e->h_fold.ptr + k->job_lo, (uint64_t*)e->d_fold.ptr + k->job_lo, sizeof(uint64_t) * nj, hipMemcpyDeviceToHost, e->stream3));
  } else {
    HIPCHK(hipEventRecord(ev[EV_TAIL_E], e->stream3));
  }
  HIPCHK(hipEventRecord(ev[EV_TAIL_DONE], e->stream3));
  return 0;
}

/* Rice initial parameters from the folded sums, once every tail kernel has landed */
static void finish_rice(struct SLAEncoder* e, const actx_t* a)
{
  const uint32_t C = e->wave_format.num_channels;
  const uint64_t* fold = (const uint64_t*)e->h_fold.ptr;
  uint32_t j;
  for (j = 0; j < a->njobs; j++) {
    if (e->device_ltm) {
      /* the jobs cover every non-silent block; what the device solved for a block that turned out RAW is dropped */
      const sla_hip_tail_job* jb = (const sla_hip_tail_job*)e->h_jobs.ptr + j;
      blkch_t* bc = &e->bc[(size_t)a->job_blk[j] * C + a->job_ch[j]];
      uint32_t t;
      if (e->blk[a->job_blk[j]].type != SLAI_BLK_COMPRESS) { continue; }
      bc->pitch = jb->pitch;
      for (t = 0; t < SLAI_MAX_TAPS; t++) { bc->ltm_q[t] = jb->ltm_coef[t]; }
    }
    {
    /* mean of the folded residual, at least 1                 src/SLACoder.c:371-384 */
    const uint64_t mean = fold[j] / e->blk[a->job_blk[j]].nsmpl;
    const uint32_t init = (uint32_t)(mean > 1 ? mean : 1);
    /* the coder keeps the parameter as a 32-bit 24.8 fixed-point word (src/SLACoder.c:14,19):
     * store the value that survives that round trip, which is also what is transmitted */
    const uint32_t kept = (uint32_t)((((uint64_t)(uint32_t)(init << 8)) + 128u) >> 8);
    e->bc[(size_t)a->job_blk[j] * C + a->job_ch[j]].rice_init = kept ? kept : 1u;
    }
  }
}

/* wait for k_expand's counts of one chunk: the kernel writes them into page-locked memory, the sequence word last, so the
 * host sees them a few microseconds after the kernel's last store instead of an event's completion latency later.  The
 * event is the safety net (a platform whose host memory the device writes through a cache the CPU does not snoop). */
static int wait_counts(volatile const uint32_t* hc, uint32_t seq, hipEvent_t done)
{
  uint32_t spins = 0;
  for (;;) {
    if (hc[3] == seq) { __sync_synchronize(); return 0; }
    if ((++spins & 1023u) == 0) {
      const hipError_t q = hipEventQuery(done);
      if (q == hipSuccess) { __sync_synchronize(); return (hc[3] == seq) ? 0 : SLA_APIRESULT_NG; }
      if (q != hipErrorNotReady) { return SLA_APIRESULT_NG; }
    }
    __asm__ volatile("" ::: "memory");
  }
}

static float ev_ms(hipEvent_t s, hipEvent_t t) { float ms = 0.f; return (hipEventElapsedTime(&ms, s, t) == hipSuccess) ? ms : 0.f; }

/* spans, chunk cuts and the search kernels of every chunk (+ k_plan, k_expand and the copies behind them).  Called once
 * the prepass result is known -- or, on a guess, while the prepass is still running (pipeline_prepare). */
static int launch_searches(struct SLAEncoder* e, actx_t* a, int preset_blocks, int trace, double t_begin)
{
  const uint32_t C = e->wave_format.num_channels;
  uint32_t c, i, want_chunks;
  int rc = 0;
  e->expand_seq += 1; if (e->expand_seq == 0) { e->expand_seq = 1; }
  memset(a->ck, 0, sizeof(a->ck));
  a->prelaunched = 0;                              /* (a second call after a wrong guess starts from scratch) */
  /* the span slots are cleared on the search stream: every kernel that writes one is launched after the host has seen a
   * search of this run complete, i.e. behind this memset */
  /* (with the tile-sum search those words, the rerun counter and k_expand's running numbers are cleared by the first search
   * kernel itself -- sla_hip_launch_extra.clear_ptr -- instead of by three fill kernels in front of it) */
  a->clear_in_kernel = (!preset_blocks && a->exact);
  if (dev_reserve(&e->d_spans, sizeof(unsigned long long) * MAX_CHUNKS * 4 * 2) != 0
      || (!a->clear_in_kernel && hipMemsetAsync(e->d_spans.ptr, 0, sizeof(unsigned long long) * MAX_CHUNKS * 4 * 2, e->stream) != hipSuccess)
      || (preset_blocks && hipStreamSynchronize(e->stream) != hipSuccess)) { return SLA_APIRESULT_NG; }
  TRACE("reserved", 0);

  /* chunking: equal runs of super-frames */
  want_chunks = e->chunks;
  /* chunks pay off once the kernels are throughput-bound; a short file (a 10-second clip: 118 super-frames) is one
   * block's serial LMS / Rice chain per stage however it is cut, and every extra chunk adds one more of those */
  if (preset_blocks || (a->nsf < 1024 && !e->chunks_forced)) { want_chunks = 1; }
  /* Chunks exist to plan one part of the file on the host while the device works on another.  With device-written block
   * tables nothing waits for the host, and one chunk saves the second set of launches and of half-filled kernels
   * (tests/tools/chunk_sweep.py: C2 1.57 -> 1.39 ms, C3-600 s 2.89 -> 2.82, C5-120 s 5.15 -> 5.05) */
  if (a->expand && !e->chunks_forced) { want_chunks = 1; }
  if (want_chunks > a->nsf / 32 + 1) { want_chunks = a->nsf / 32 + 1; }
  if (want_chunks > MAX_CHUNKS) { want_chunks = MAX_CHUNKS; }
  if (want_chunks < 1) { want_chunks = 1; }
  a->nchunks = want_chunks;
  /* the block stages of the two chunks on two streams, so that one chunk's lattice / FFT kernels fill the gaps of the
   * other's block stage.  Round 2 kept this to big files (k_lpc_blocks of a ten-minute mono file was too short to share
   * the device: 2.32 -> 2.43 ms); with the certified block stage of round 3 the ten-minute mono file gains the most
   * (1.61 -> 1.52 ms), C3-600 s 3.03 -> 2.96, C5-120 s 5.26 -> 5.33 (within the run-to-run spread): on whenever chunked */
  e->alt_now = (e->alt_streams != 0);
  if (!(e->device_ltm && e->single_tail) || a->nchunks < 2) { e->alt_now = 0; }
  a->one_stream = (e->one_stream && a->expand && a->nchunks == 1 && e->device_ltm && e->single_tail);
  /* chunk boundaries in 1/1000 of the super-frames: equal parts unless SLA_HIP_CHUNK_SPLIT gave the shares */
  if (e->first_chunk != 0 && a->nchunks >= 2) {
    /* option first_chunk: that many 1/1000 of the super-frames in chunk 0, the others share the rest equally */
    e->chunk_cut[0] = 0;
    for (c = 1; c <= a->nchunks; c++) { e->chunk_cut[c] = e->first_chunk + (1000u - e->first_chunk) * (c - 1) / (a->nchunks - 1); }
  } else if (e->split_count != a->nchunks && a->nchunks == 2) {
    /* measured on C2, C3-600 s, C5-120 s (tests/tools/chunk_sweep.py, medians of interleaved rounds): two chunks, the
     * first one short -- the host plans it while nothing else can run, and plans the second under the first one's
     * kernels.  With one k_tail for the file (device long-term solve) 25 % / 75 % is best: 2.32 / 5.94 / 11.97 ms
     * against 2.46 / 6.23 / 12.24 ms in one chunk; a third chunk only adds launches.  With a k_tail per chunk
     * (host solve) every chunk costs one more serial LMS chain, and 40 % / 60 % was the best cut. */
    e->chunk_cut[0] = 0; e->chunk_cut[1] = (e->device_ltm && e->single_tail && !e->alt_now) ? 250 : 400; e->chunk_cut[2] = 1000;
  } else if (e->split_count != a->nchunks && a->nchunks == 3) {
    e->chunk_cut[0] = 0; e->chunk_cut[1] = 250; e->chunk_cut[2] = 625; e->chunk_cut[3] = 1000;
  } else if (e->split_count != a->nchunks) {
    for (c = 0; c <= a->nchunks; c++) { e->chunk_cut[c] = 1000u * c / a->nchunks; }
  } else {
    uint32_t acc = 0, sum = 0;
    for (c = 0; c < a->nchunks; c++) { sum += e->split[c]; }
    for (c = 0; c <= a->nchunks; c++) { e->chunk_cut[c] = 1000u * acc / sum; if (c < a->nchunks) { acc += e->split[c]; } }
  }
  for (c = 0; c < a->nchunks; c++) {
    chunk_t* k = &a->ck[c];
    k->sf_lo = (uint32_t)((uint64_t)a->nsf * e->chunk_cut[c] / 1000);
    k->sf_hi = (c + 1 == a->nchunks) ? a->nsf : (uint32_t)((uint64_t)a->nsf * e->chunk_cut[c + 1] / 1000);
    if (!preset_blocks && k->sf_hi > k->sf_lo) {
      uint32_t last_live = 0xFFFFFFFFu, first_live = 0xFFFFFFFFu;
      k->grp_lo = a->sf[k->sf_lo].grp_lo; k->grp_hi = a->sf[k->sf_hi - 1].grp_hi;
      for (i = k->sf_lo; i < k->sf_hi; i++) { if (a->sf[i].shape != 0xFFFFFFFFu) { if (first_live == 0xFFFFFFFFu) { first_live = i; } last_live = i; } }
      if (first_live != 0xFFFFFFFFu) {
        k->slot_lo = a->sf[first_live].slot_base;
        k->slot_hi = a->sf[last_live].slot_base + C * a->shapes[a->sf[last_live].shape].ncand;
        k->xg_lo = a->sf[first_live].xg; k->xg_hi = a->sf[last_live].xg + C;
      }
    }
  }

  if (!preset_blocks) {
    if (a->clear_in_kernel) {
      /* (the first search launch of this analysis takes them along: search_launch, chunk 0) */
      a->clear_ptr[0] = (uint32_t*)e->d_spans.ptr; a->clear_words[0] = MAX_CHUNKS * 4 * 2 * 2;
      a->clear_ptr[1] = (uint32_t*)e->d_or.ptr + 2; a->clear_words[1] = 1;
      a->clear_ptr[2] = a->expand ? (uint32_t*)e->d_run.ptr : NULL; a->clear_words[2] = 4;
    } else if (hipMemsetAsync((uint32_t*)e->d_or.ptr + 2, 0, sizeof(uint32_t), e->stream) != hipSuccess) { rc = SLA_APIRESULT_NG; }      /* groups rerun as serial chains */
    for (c = 0; c < a->nchunks && rc == 0; c++) { rc = search_launch(e, a, c); }
    TRACE("search launched", a->nchunks);
  } else {
    a->ck[0].blk_lo = 0; a->ck[0].blk_hi = e->num_blocks;
  }
  return rc;
}

/* run the pipeline.  preset_blocks != 0: the block table is already in e->blk (EncodeBlock), no search. */
/* SLA_HIP_TRACE=1: host-side timeline of one analysis on stderr (ms since the start of run_pipeline) */

static int run_pipeline(struct SLAEncoder* e, int preset_blocks)
{
  const int trace = e->trace;
  const double t_begin = now_ms();
  const uint32_t C = e->wave_format.num_channels;
  actx_t a;
  uint32_t c, i;
  double t_host = 0.0, t0;
  int rc = 0, tail_queued = 0;
  memset(&a, 0, sizeof(a));
  a.ev = e->ev + 2;

  e->fallback_groups = 0; e->host_planned = 0; e->blocks_exact = 0; e->cert_broken = 0; e->audit_ok = 0; e->audit_bad = 0;
  e->expanded_chunks = 0;
  e->cert_now = (e->block_cert && !(e->fuse_lattice && e->encode_param.parcor_order <= 64) && !e->tune.lpc_blocks_chains
                 && sla_hip_search_exact_lags(e->encode_param.parcor_order) != 0);
  a.trace = trace; a.t_begin = t_begin;
  if (!preset_blocks) {
    e->num_blocks = 0;
    if ((rc = pipeline_prepare(e, &a)) != 0) { actx_free(&a); return rc; }
  } else {
    extern uint32_t sla_hip_lattice_chunk_samples(uint32_t order);
    const uint32_t cs = sla_hip_lattice_chunk_samples(e->encode_param.parcor_order);
    uint32_t woff;
    a.blocks_bound = e->num_blocks;
    for (i = 0; i < e->num_blocks; i++) {
      a.lchunks_bound += C * (e->blk[i].nsmpl / cs + 2);
      if (e->blk[i].type != SLAI_BLK_SILENT && (rc = window_offset(e, e->blk[i].nsmpl, &woff)) != 0) { actx_free(&a); return rc; }
    }
    if ((rc = pin_reserve(&e->h_cands, 64)) != 0 || (rc = pin_reserve(&e->h_groups, 64)) != 0) { actx_free(&a); return rc; }
  }
  {
    const size_t nslots = (size_t)a.blocks_bound * C + 1;
    a.job_blk = (uint32_t*)malloc(sizeof(uint32_t) * nslots);
    a.job_ch = (uint32_t*)malloc(sizeof(uint32_t) * nslots);
    a.job_grp = (uint32_t*)malloc(sizeof(uint32_t) * nslots);
    a.grp_of_slot = (uint32_t*)malloc(sizeof(uint32_t) * nslots);
    a.parts = (uint32_t*)malloc(sizeof(uint32_t) * ((size_t)a.nsf + 1) * SLAI_MAX_NODES);
    a.nparts = (uint32_t*)calloc((size_t)a.nsf + 1, sizeof(uint32_t));
    a.status = (int*)calloc((size_t)a.nsf + 1, sizeof(int));
    if (!a.job_blk || !a.job_ch || !a.job_grp || !a.grp_of_slot || !a.parts || !a.nparts || !a.status) { actx_free(&a); return SLA_APIRESULT_NG; }
  }
  TRACE("prepared (prepass + tables)", a.nsf);
  if (preset_blocks && (rc = pipeline_reserve(e, &a)) != 0) { actx_free(&a); return rc; }      /* (otherwise done while the prepass ran) */
  if (!a.spec && (rc = launch_searches(e, &a, preset_blocks, trace, t_begin)) != 0) { actx_free(&a); return rc; }
  /* device-written block tables: every chunk's block stage goes out as soon as its two counts are in, then the one tail
   * -- the device has its whole queue before the host starts on its own copy of the tables.  A chunk the device could
   * not finish (a partition that needs the host's logarithms) ends this: it and the chunks behind it take the host route. */
  if (!preset_blocks && a.expand && rc == 0) {
    for (c = 0; c < a.nchunks && rc == 0; c++) {
      volatile const uint32_t* hc = (volatile const uint32_t*)e->h_counts.ptr + 4 * (size_t)c;
      if (a.ck[c].grp_hi == a.ck[c].grp_lo) { break; }      /* nothing searched in this chunk: no k_expand ran */
      if ((rc = wait_counts(hc, e->expand_seq, a.ev[(size_t)c * EV_PER_CHUNK + EV_EXPANDED])) != 0) { break; }
      if (hc[2] != 1u) { break; }
      a.dev_lo[c][0] = a.dev_blk; a.dev_lo[c][1] = a.dev_blk + hc[0]; a.dev_lo[c][2] = a.dev_bg; a.dev_lo[c][3] = a.dev_bg + hc[1];
      a.dev_blk += hc[0]; a.dev_bg += hc[1];
      if (a.dev_blk > a.blocks_bound || (size_t)a.dev_bg > (size_t)a.blocks_bound * C) { rc = SLA_APIRESULT_NG; break; }
      TRACE("counts in", c);
      if ((rc = blocks_launch(e, &a, c, 1)) != 0) { break; }
      a.launched[c] = 1; e->expanded_chunks++;
      TRACE("blocks launched (device tables)", c);
    }
    if (rc == 0 && e->expanded_chunks == a.nchunks && e->device_ltm && e->single_tail) {
      rc = tail_enqueue(e, &a, a.nchunks - 1, 0, a.dev_bg);
      tail_queued = 1;
      TRACE("tail queued", a.nchunks - 1);
    }
  }
  /* software pipeline over chunks: plan(c) | blocks(c) ; solve+tail(c-1) */
  for (c = 0; c < a.nchunks && rc == 0; c++) {
    hipEvent_t* ev = a.ev + (size_t)c * EV_PER_CHUNK;
    if (!preset_blocks) {
      if (hipEventSynchronize(ev[EV_SEARCH_DONE]) != hipSuccess) { rc = SLA_APIRESULT_NG; break; }
      TRACE("search done", c);
      t0 = now_ms();
      rc = plan_chunk(e, &a, c);
      t_host += now_ms() - t0;
      if (rc != 0) { break; }
      TRACE("planned", c);
    }
    if ((rc = blocks_launch(e, &a, c, a.launched[c] ? 2 : 0)) != 0) { break; }
    TRACE(a.launched[c] ? "host tables" : "blocks launched", c);
    if (c >= 1) {
      hipEvent_t* pv = a.ev + (size_t)(c - 1) * EV_PER_CHUNK;
      if (hipEventSynchronize(pv[EV_LPC_DOWN]) != hipSuccess) { rc = SLA_APIRESULT_NG; break; }
      if ((rc = raw_phase(e, &a, c - 1)) != 0) { break; }
      if (e->device_ltm) { continue; }
      if (hipEventSynchronize(pv[EV_BLOCK_DONE]) != hipSuccess) { rc = SLA_APIRESULT_NG; break; }
      TRACE("blocks done", c - 1);
      if (e->single_tail) {
        t0 = now_ms();
        ltm_solve(e, &a, a.ck[c - 1].job_hi);      /* under the kernels of chunk c */
        e->timing[6] += (float)(now_ms() - t0);
      } else {
        t0 = now_ms();
        rc = tail_launch(e, &a, c - 1);
        e->timing[6] += (float)(now_ms() - t0);
        TRACE("tail launched", c - 1);
      }
    }
  }
  if (rc == 0 && e->device_ltm) {
    /* everything is queued: one k_tail over all groups (unless every chunk queued its own), then the RAW decision of
     * the last chunk on the host threads while the device works */
    hipEvent_t* pv = a.ev + (size_t)(a.nchunks - 1) * EV_PER_CHUNK;
    if (e->single_tail && !tail_queued) { rc = tail_enqueue(e, &a, a.nchunks - 1, 0, a.nbg); TRACE("tail queued", a.nchunks - 1); }
    if (tail_queued && a.nbg != a.dev_bg) { rc = SLA_APIRESULT_NG; }
    if (rc == 0 && hipEventSynchronize(pv[EV_LPC_DOWN]) != hipSuccess) { rc = SLA_APIRESULT_NG; }
    if (rc == 0) { rc = raw_phase(e, &a, a.nchunks - 1); }
    TRACE("RAW decided", a.nchunks - 1);
  } else if (rc == 0) {
    hipEvent_t* pv = a.ev + (size_t)(a.nchunks - 1) * EV_PER_CHUNK;
    if (hipEventSynchronize(pv[EV_LPC_DOWN]) != hipSuccess || (rc = raw_phase(e, &a, a.nchunks - 1)) != 0
        || hipEventSynchronize(pv[EV_BLOCK_DONE]) != hipSuccess) { if (rc == 0) { rc = SLA_APIRESULT_NG; } }
    else {
      /* one k_tail for the whole file: its duration is one block's serial LMS chain however many blocks run beside it, so
       * the chunks share it (the BLOCK_DONE events sit in order on the download stream: the last one covers them all) */
      if (e->single_tail) { a.ck[a.nchunks - 1].job_lo = a.ck[0].job_lo; }
      TRACE("blocks done", a.nchunks - 1);
      t0 = now_ms();
      rc = tail_launch(e, &a, a.nchunks - 1);
      e->timing[6] += (float)(now_ms() - t0);
      TRACE("tail launched", a.nchunks - 1);
    }
  }
  /* the rerun counter and the kernels' execution spans ride home behind the last k_tail (every other stream has been
   * waited for by then): two synchronous copies here cost 70-90 us per step */
  {
    unsigned long long* sp_host = (unsigned long long*)((uint8_t*)e->h_or + 64);
    int copied = 0;
    if (rc == 0) {
      hipStream_t last = (a.tail_stream != NULL) ? a.tail_stream : ((e->device_ltm && e->single_tail) ? e->stream2 : e->stream3);      /* the stream the last k_tail runs on */
      copied = (hipMemcpyAsync(sp_host, e->d_spans.ptr, sizeof(unsigned long long) * MAX_CHUNKS * 4 * 2, hipMemcpyDeviceToHost, last) == hipSuccess);
      if (!preset_blocks && hipMemcpyAsync(e->h_or + 2, (uint32_t*)e->d_or.ptr + 2, sizeof(uint32_t), hipMemcpyDeviceToHost, last) != hipSuccess) { rc = SLA_APIRESULT_NG; }
    }
  if (hipStreamSynchronize(e->stream) != hipSuccess || hipStreamSynchronize(e->stream2) != hipSuccess
      || hipStreamSynchronize(e->stream_up) != hipSuccess || hipStreamSynchronize(e->stream_down) != hipSuccess
      || hipStreamSynchronize(e->stream3) != hipSuccess) { if (rc == 0) { rc = SLA_APIRESULT_NG; } }
  if (rc == 0 && !preset_blocks) { e->fallback_groups = e->h_or[2]; }
  if (rc == 0 && e->cert_now) {
    const uint32_t* cnt = (const uint32_t*)e->h_cert_flag.ptr + (size_t)a.blocks_bound * C + 1;
    for (c = 0; c < a.nchunks; c++) { if (a.ck[c].bg_hi > a.ck[c].bg_lo) { e->blocks_exact += cnt[c]; } }
    /* the list the exact kernels walked holds the uncertified pairs and the audited ones: report the former */
    e->blocks_exact -= (e->audit_ok + e->audit_bad <= e->blocks_exact) ? (e->audit_ok + e->audit_bad) : e->blocks_exact;
    if (e->cert_broken) { rc = SLA_APIRESULT_NG; }
  }
  memset(e->kernel_ms, 0, sizeof(e->kernel_ms));
  if (rc == 0) {
    const unsigned long long* sp = sp_host;
    int rate_khz = e->wall_clock_khz;
    if (rate_khz <= 0) {
      rate_khz = 100000;
      (void)hipDeviceGetAttribute(&rate_khz, hipDeviceAttributeWallClockRate, e->device);
      if (rate_khz <= 0) { rate_khz = 100000; }
      e->wall_clock_khz = rate_khz;
    }
    if (copied) {
      for (c = 0; c < a.nchunks; c++) {
        for (i = 0; i < 4; i++) {
          const unsigned long long st = ~sp[(c * 4 + i) * 2], en = sp[(c * 4 + i) * 2 + 1];
          if (en != 0 && en > st) { e->kernel_ms[i] += (float)((double)(en - st) / (double)rate_khz); }
        }
      }
    }
  }
  }
  TRACE("all streams idle", 0);
  if (rc == 0) {
    finish_rice(e, &a);
    e->timing[0] = preset_blocks ? 0.f : ev_ms(e->ev[0], e->ev[1]);
    e->timing[5] = (float)t_host;
    for (c = 0; c < a.nchunks; c++) {
      hipEvent_t* ev = a.ev + (size_t)c * EV_PER_CHUNK;
      if (!preset_blocks) { e->timing[1] += ev_ms(ev[EV_SEARCH_S], ev[EV_SEARCH_E]); }
      e->timing[2] += ev_ms(ev[EV_LPCB_S], ev[EV_LPCB_E]);
      e->timing[3] += ev_ms(ev[EV_LPCB_E], ev[EV_LAT_E]);
      e->timing[8] += ev_ms(ev[EV_ACF_S], ev[EV_ACF_E]);
      if (!e->single_tail || c == a.nchunks - 1) { e->timing[4] += ev_ms(ev[EV_TAIL_S], ev[EV_TAIL_E]); }
    }
    e->timing[9] = (float)a.nchunks;
    e->tail_launches = (e->single_tail) ? 1u : a.nchunks;
    e->timing[10] = (float)e->fallback_groups;
    e->timing[11] = (float)a.exact;
  }
  actx_free(&a);
  return rc;
}

/* every entry point that touches the GPU: the handle's device becomes the calling thread's current device (a handle
 * is bound to the device that was current at SLAEncoder_Create; a multi-GPU process holds one handle per device) and
 * the handle's launcher knobs are the ones the launchers of this thread see */
static int enter(struct SLAEncoder* e)
{
  int cur = -1;
  if (hipGetDevice(&cur) != hipSuccess || cur != e->device) { HIPCHK(hipSetDevice(e->device)); }
  sla_hip_use_tuning(&e->tune);
  return 0;
}

int sla_hip_encoder_set_option(struct SLAEncoder* e, const char* name, double value)
{
  const long iv = (long)value;
  if (e == NULL || name == NULL || !(value == value)) { return SLA_APIRESULT_INVALID_ARGUMENT; }
#define OPT_RANGE(lo, hi) do { if (value != (double)iv || iv < (lo) || iv > (hi)) { return SLA_APIRESULT_INVALID_ARGUMENT; } } while (0)
  if (strcmp(name, "lpc_pack") == 0)               { OPT_RANGE(0, 4); e->tune.lpc_pack = (uint32_t)iv; }
  else if (strcmp(name, "lpc_threads") == 0)       { if (iv != 0 && iv != 256 && iv != 512) { return SLA_APIRESULT_INVALID_ARGUMENT; } e->tune.lpc_threads = (uint32_t)iv; }
  else if (strcmp(name, "tail_waves") == 0)        { OPT_RANGE(0, 4); e->tune.tail_waves = (uint32_t)iv; }
  else if (strcmp(name, "lpc_tile") == 0)          { if (iv != 0 && iv != 24 && iv != 48) { return SLA_APIRESULT_INVALID_ARGUMENT; } e->tune.lpc_tile = (uint32_t)iv; }
  else if (strcmp(name, "rice_lanes") == 0)        { OPT_RANGE(0, 2); e->tune.rice_lanes = (uint32_t)iv; }
  else if (strcmp(name, "lattice_plain") == 0)     { OPT_RANGE(0, 1); e->tune.lattice_plain = (uint32_t)iv; }
  else if (strcmp(name, "cert_audit") == 0)        { OPT_RANGE(0, 1 << 30); e->tune.cert_audit = (uint32_t)iv; }
  else if (strcmp(name, "tail_taps") == 0)         { if (iv != 0 && iv != 1 && iv != 2 && iv != 4) { return SLA_APIRESULT_INVALID_ARGUMENT; } e->tune.tail_taps = (uint32_t)iv; }
  else if (strcmp(name, "lpc_blocks_chains") == 0) { OPT_RANGE(0, 1); e->tune.lpc_blocks_chains = (uint32_t)iv; if (iv) { e->fuse_lattice = 0; } }
  /* the certification margins may only be widened: below the built-in values byte-identity is no longer guaranteed */
  else if (strcmp(name, "plan_margin") == 0)       { if (value != 0.0 && !(value >= 1e-4)) { return SLA_APIRESULT_INVALID_ARGUMENT; } e->tune.plan_margin = value; }
  else if (strcmp(name, "chunks") == 0)            { OPT_RANGE(1, 8); e->chunks = (uint32_t)iv; e->split_count = 0; e->chunks_forced = 1; }
  else if (strcmp(name, "search_exact") == 0)      { OPT_RANGE(0, 1); e->search_exact = (int)iv; }
  else if (strcmp(name, "exact_bits") == 0)        { OPT_RANGE(1, 53); e->exact_bits = (int)iv; }
  else if (strcmp(name, "cert_safety") == 0)       { if ((value != 0.0 && value < 64.0) || value > 1e30) { return SLA_APIRESULT_INVALID_ARGUMENT; } e->cert_safety = value; }
  else if (strcmp(name, "upload24") == 0)          { OPT_RANGE(0, 1); e->upload24 = (int)iv; }
  else if (strcmp(name, "block_cert") == 0)        { OPT_RANGE(0, 1); e->block_cert = (int)iv; }
  else if (strcmp(name, "block_cert_safety") == 0) { if (!(value >= 16.0) || value > 1e30) { return SLA_APIRESULT_INVALID_ARGUMENT; } e->block_cert_safety = value; }
  else if (strcmp(name, "device_plan") == 0)       { OPT_RANGE(0, 1); e->device_plan = (int)iv; }
  else if (strcmp(name, "single_tail") == 0)       { OPT_RANGE(0, 1); e->single_tail = (int)iv; }
  else if (strcmp(name, "stream") == 0)            { OPT_RANGE(0, 1); e->stream_mode = (int)iv; }
  else if (strcmp(name, "stream_piece") == 0)      { OPT_RANGE(1024, 1 << 30); e->stream_piece = (uint32_t)iv; }
  else if (strcmp(name, "batch_lanes") == 0)       { OPT_RANGE(1, SLAI_STREAM_LANES); e->batch_lanes = (uint32_t)iv; }
  else if (strcmp(name, "stream_lanes") == 0)      { OPT_RANGE(1, SLAI_STREAM_LANES); e->stream_lanes = (uint32_t)iv; }
  else if (strcmp(name, "first_chunk") == 0)       { OPT_RANGE(0, 999); e->first_chunk = (uint32_t)iv; }
  else if (strcmp(name, "alt_streams") == 0)       { OPT_RANGE(0, 2); e->alt_streams = (int)iv; }
  else if (strcmp(name, "device_expand") == 0)     { OPT_RANGE(0, 1); e->device_expand = (int)iv; }
  else if (strcmp(name, "expand_silence") == 0)    { OPT_RANGE(0, 1); e->expand_silence = (int)iv; }
  else if (strcmp(name, "one_stream") == 0)        { OPT_RANGE(0, 1); e->one_stream = (int)iv; }
  else if (strcmp(name, "prelaunch") == 0)         { OPT_RANGE(0, 1); e->prelaunch = (int)iv; }
  else if (strcmp(name, "table_cache") == 0)       { OPT_RANGE(0, 1); e->table_cache = (int)iv; e->tab_valid = 0; e->spec_valid = 0; }
  else if (strcmp(name, "device_ltm") == 0)        { OPT_RANGE(0, 1); e->device_ltm = (int)iv; }
  else if (strcmp(name, "fuse_lattice") == 0)      { OPT_RANGE(0, 1); e->fuse_lattice = (int)iv && !e->tune.lpc_blocks_chains; }
  else if (strcmp(name, "threads") == 0) {
    OPT_RANGE(1, 64);
    if ((uint32_t)iv != e->threads) {
      struct slai_pool* np = pool_create((uint32_t)iv);
      if (np == NULL) { return SLA_APIRESULT_NG; }
      pool_destroy(e->pool); e->pool = np; e->threads = (uint32_t)iv;
    }
  }
  else { return SLA_APIRESULT_INVALID_ARGUMENT; }
#undef OPT_RANGE
  e->analysed = 0;
  return 0;
}

static int check_ready(const struct SLAEncoder* e)
{
  if (!(e->status_flag & STATUS_WAVE_FORMAT) || !(e->status_flag & STATUS_ENCODE_PARAM)) { return SLA_APIRESULT_PARAMETER_NOT_SET; }
  if (e->encode_param.ch_process_method == SLA_CHPROCESSMETHOD_STEREO_MS && e->wave_format.num_channels != 2) {
    return SLA_APIRESULT_INVAILD_CHPROCESSMETHOD;
  }
  if ((int)e->encode_param.window_function_type < 0
      || (int)e->encode_param.window_function_type > (int)SLA_WINDOWFUNCTIONTYPE_VORBIS) { return SLA_APIRESULT_INVALID_WINDOWFUNCTION_TYPE; }
  {
    const uint32_t l = e->encode_param.lms_order_per_filter;
    if (!(l == 4 || l == 8 || l == 16 || l == 32)) { return SLA_APIRESULT_EXCEED_HANDLE_CAPACITY; }
    if (!(e->encode_param.longterm_order & 1u)) { return SLA_APIRESULT_INVALID_ARGUMENT; }
  }
  return 0;
}

int sla_hip_analyze_device(struct SLAEncoder* e, const int32_t* d_pcm, uint64_t plane_stride,
                           uint32_t num_samples, sla_hip_stream_t stream, float* timing_ms)
{
  const double t_start = now_ms();
  int rc;
  if (e == NULL || d_pcm == NULL) { return SLA_APIRESULT_INVALID_ARGUMENT; }
  RCCHK(check_ready(e));
  if (plane_stride < num_samples) { return SLA_APIRESULT_INVALID_ARGUMENT; }
  RCCHK(enter(e));
  if (stream != NULL) { HIPCHK(hipStreamSynchronize((hipStream_t)stream)); }   /* producer of d_pcm */
  e->analysed = 0;
  e->pcm_dev = d_pcm; e->stride = plane_stride; e->num_samples = num_samples;
  memset(e->timing, 0, sizeof(e->timing));
  rc = run_pipeline(e, 0);
  if (rc != 0) { return rc; }
  e->wave_format.offset_lshift = (uint8_t)e->lshift;
  e->timing[7] = (float)(now_ms() - t_start);
  if (timing_ms != NULL) { memcpy(timing_ms, e->timing, sizeof(e->timing)); }
  e->analysed = 1;
  return 0;
}

/* ---- one file over several GPUs (include/sla_hip.h: "one file, several GPUs") --------------------------------- */

int sla_hip_shard_scan(struct SLAEncoder* e, const int32_t* d_pcm, uint64_t plane_stride, uint32_t num_samples,
                       uint32_t* or_word, uint64_t* nz_mask)
{
  const uint64_t nwords = ((uint64_t)num_samples + 63) / 64;
  uint32_t ms;
  if (e == NULL || d_pcm == NULL || or_word == NULL || (nz_mask == NULL && num_samples != 0)) { return SLA_APIRESULT_INVALID_ARGUMENT; }
  RCCHK(check_ready(e));
  if (plane_stride < num_samples) { return SLA_APIRESULT_INVALID_ARGUMENT; }
  *or_word = 0;
  if (num_samples == 0) { return 0; }
  RCCHK(enter(e));
  ms = (e->encode_param.ch_process_method == SLA_CHPROCESSMETHOD_STEREO_MS);
  RCCHK(dev_reserve(&e->d_or, 64));
  RCCHK(dev_reserve(&e->d_nz, (size_t)(nwords + 2) * 8));
  RCCHK(pin_reserve(&e->h_nz, (size_t)(nwords + 2) * 8));
  RCCHK(sla_hip_launch_prepass(d_pcm, plane_stride, e->wave_format.num_channels, num_samples, e->wave_format.bit_per_sample, ms,
                               (uint32_t*)e->d_or.ptr, (uint64_t*)e->d_nz.ptr, e->stream));
  HIPCHK(hipMemcpyAsync(e->h_or, e->d_or.ptr, 8, hipMemcpyDeviceToHost, e->stream));
  HIPCHK(hipMemcpyAsync(e->h_nz.ptr, e->d_nz.ptr, (size_t)nwords * 8, hipMemcpyDeviceToHost, e->stream));
  HIPCHK(hipStreamSynchronize(e->stream));
  e->nz_ones_words = 0;                                   /* h_nz no longer holds what the single-file path remembers */
  *or_word = e->h_or[0];
  memcpy(nz_mask, e->h_nz.ptr, (size_t)nwords * 8);
  return 0;
}

/* The cheap form of the scan: OR word and the number of all-zero 64-sample mask words of the piece, 8 bytes home
 * instead of num_samples / 8.  A silence run only moves a super-frame start when it is at least SLAI_MIN_BLOCK (2048)
 * samples long, i.e. contains all-zero mask words: when no rank counts any, sla_hip_shard_bounds needs no mask at all
 * (nz_mask = NULL) and the ranks exchange 12 bytes each instead of the mask -- the usual case. */
int sla_hip_shard_scan_counts(struct SLAEncoder* e, const int32_t* d_pcm, uint64_t plane_stride, uint32_t num_samples,
                              uint32_t* or_word, uint32_t* zero_mask_words)
{
  const uint64_t nwords = ((uint64_t)num_samples + 63) / 64;
  uint32_t ms;
  if (e == NULL || d_pcm == NULL || or_word == NULL || zero_mask_words == NULL) { return SLA_APIRESULT_INVALID_ARGUMENT; }
  RCCHK(check_ready(e));
  if (plane_stride < num_samples) { return SLA_APIRESULT_INVALID_ARGUMENT; }
  *or_word = 0; *zero_mask_words = 0;
  if (num_samples == 0) { return 0; }
  RCCHK(enter(e));
  ms = (e->encode_param.ch_process_method == SLA_CHPROCESSMETHOD_STEREO_MS);
  RCCHK(dev_reserve(&e->d_or, 64));
  RCCHK(dev_reserve(&e->d_nz, (size_t)(nwords + 2) * 8));
  RCCHK(sla_hip_launch_prepass(d_pcm, plane_stride, e->wave_format.num_channels, num_samples, e->wave_format.bit_per_sample, ms,
                               (uint32_t*)e->d_or.ptr, (uint64_t*)e->d_nz.ptr, e->stream));
  HIPCHK(hipMemcpyAsync(e->h_or, e->d_or.ptr, 8, hipMemcpyDeviceToHost, e->stream));
  HIPCHK(hipStreamSynchronize(e->stream));
  *or_word = e->h_or[0];
  *zero_mask_words = e->h_or[1];
  return 0;
}

/* The super-frame hop of the whole file (src/SLAEncoder.c:846-869 with the silence shortcut of :392-408), on the
 * host from the 1-bit mask; bounds[r] = the first super-frame start at or behind sample r * N / world. */
int sla_hip_shard_bounds(uint32_t num_samples, uint32_t max_num_block_samples, const uint64_t* nz_mask, uint32_t world,
                         uint32_t* bounds)
{
  uint32_t pos = 0, r = 1;
  /* nz_mask == NULL: no rank saw an all-zero mask word (sla_hip_shard_scan_counts), nothing is silent */
  if (bounds == NULL || world == 0 || max_num_block_samples < SLAI_MIN_BLOCK) { return SLA_APIRESULT_INVALID_ARGUMENT; }
  bounds[0] = 0;
  if (nz_mask == NULL) {
    /* no silence: the hop visits every multiple of the block length -- the first one at or behind each target */
    for (r = 1; r < world; r++) {
      const uint64_t target = ((uint64_t)num_samples * r + world - 1) / world;
      const uint64_t at = (target + max_num_block_samples - 1) / max_num_block_samples * max_num_block_samples;
      bounds[r] = (at < num_samples) ? (uint32_t)at : num_samples;
    }
    bounds[world] = num_samples;
    return 0;
  }
  while (pos < num_samples && r < world) {
    const uint32_t remain = num_samples - pos;
    const uint32_t window = (max_num_block_samples < remain) ? max_num_block_samples : remain;
    const uint32_t min_blk = (SLAI_MIN_BLOCK < remain) ? SLAI_MIN_BLOCK : remain;
    const uint32_t run = slai_zero_run(nz_mask, pos, window);
    while (r < world && (uint64_t)pos >= ((uint64_t)num_samples * r + world - 1) / world) { bounds[r++] = pos; }
    pos += (run >= min_blk) ? run : window;
  }
  while (r <= world) { bounds[r++] = num_samples; }
  return 0;
}

int sla_hip_shard_analyze(struct SLAEncoder* e, const int32_t* d_pcm, uint64_t plane_stride, uint32_t num_samples,
                          uint32_t file_or_word, float* timing_ms)
{
  int rc;
  if (e == NULL) { return SLA_APIRESULT_INVALID_ARGUMENT; }
  e->file_or_word = file_or_word;
  rc = sla_hip_analyze_device(e, d_pcm, plane_stride, num_samples, NULL, timing_ms);
  e->file_or_word = 0;
  return rc;
}

/* sla_hip_shard_analyze for a file whose scan (sla_hip_shard_scan_counts on every rank) counted no all-zero mask word:
 * with the file's OR word known and no silence anywhere, the range needs no prepass of its own */
int sla_hip_shard_analyze_no_silence(struct SLAEncoder* e, const int32_t* d_pcm, uint64_t plane_stride, uint32_t num_samples,
                                     uint32_t file_or_word, float* timing_ms)
{
  int rc;
  if (e == NULL || file_or_word == 0) { return SLA_APIRESULT_INVALID_ARGUMENT; }
  e->file_or_word = file_or_word; e->skip_prepass = 1;
  rc = sla_hip_analyze_device(e, d_pcm, plane_stride, num_samples, NULL, timing_ms);
  e->file_or_word = 0; e->skip_prepass = 0;
  return rc;
}

int sla_hip_shard_header(const uint8_t* const* shard_headers, uint32_t world, uint8_t* data, uint32_t data_size)
{
  struct SLAHeaderInfo total, h;
  uint32_t r;
  if (shard_headers == NULL || world == 0 || data == NULL) { return SLA_APIRESULT_INVALID_ARGUMENT; }
  memset(&total, 0, sizeof(total));
  for (r = 0; r < world; r++) {
    if (shard_headers[r] == NULL) { return SLA_APIRESULT_INVALID_ARGUMENT; }
    if (SLADecoder_DecodeHeader(shard_headers[r], SLA_HEADER_SIZE, &h) != SLA_APIRESULT_OK) { return SLA_APIRESULT_INVALID_HEADER_FORMAT; }
    if (r == 0) { total = h; continue; }
    if (memcmp(&h.encode_param, &total.encode_param, sizeof(h.encode_param)) != 0 || h.wave_format.num_channels != total.wave_format.num_channels
        || h.wave_format.bit_per_sample != total.wave_format.bit_per_sample || h.wave_format.sampling_rate != total.wave_format.sampling_rate
        || h.wave_format.offset_lshift != total.wave_format.offset_lshift) { return SLA_APIRESULT_INVALID_HEADER_FORMAT; }
    if ((uint64_t)total.num_samples + h.num_samples > 0xFFFFFFFFull) { return SLA_APIRESULT_EXCEED_HANDLE_CAPACITY; }
    total.num_samples += h.num_samples; total.num_blocks += h.num_blocks;
    if (h.max_block_size > total.max_block_size) { total.max_block_size = h.max_block_size; }
    if (h.max_bit_per_second > total.max_bit_per_second) { total.max_bit_per_second = h.max_bit_per_second; }
  }
  return slai_write_header(&total, data, data_size);
}

int sla_hip_last_kernel_ms(const struct SLAEncoder* e, float* kernel_ms)
{
  if (e == NULL || kernel_ms == NULL) { return SLA_APIRESULT_INVALID_ARGUMENT; }
  memcpy(kernel_ms, e->kernel_ms, sizeof(e->kernel_ms));
  return 0;
}

int sla_hip_last_counters(const struct SLAEncoder* e, uint32_t* counters)
{
  if (e == NULL || counters == NULL) { return SLA_APIRESULT_INVALID_ARGUMENT; }
  counters[0] = e->fallback_groups; counters[1] = e->host_planned;
  counters[2] = (uint32_t)e->timing[11]; counters[3] = (uint32_t)e->device_plan;
  counters[4] = e->tail_launches; counters[5] = (uint32_t)e->device_ltm;
  return 0;
}

int sla_hip_last_expand(const struct SLAEncoder* e, uint32_t* counters)
{
  if (e == NULL || counters == NULL) { return SLA_APIRESULT_INVALID_ARGUMENT; }
  counters[0] = e->expanded_chunks; counters[1] = (uint32_t)e->timing[9]; counters[2] = e->table_hits; counters[3] = e->spec_misses;
  return 0;
}

int sla_hip_last_block_cert(const struct SLAEncoder* e, uint32_t* counters)
{
  if (e == NULL || counters == NULL) { return SLA_APIRESULT_INVALID_ARGUMENT; }
  counters[0] = (uint32_t)e->cert_now; counters[1] = e->blocks_exact;
  return 0;
}

int sla_hip_last_cert_audit(const struct SLAEncoder* e, uint32_t* counters)
{
  if (e == NULL || counters == NULL) { return SLA_APIRESULT_INVALID_ARGUMENT; }
  counters[0] = e->audit_ok; counters[1] = e->audit_bad;
  return 0;
}

int sla_hip_last_timing(const struct SLAEncoder* e, float* timing_ms)
{
  if (e == NULL || timing_ms == NULL) { return SLA_APIRESULT_INVALID_ARGUMENT; }
  memcpy(timing_ms, e->timing, sizeof(e->timing));
  return 0;
}

/* ------------------------------------------------------------------------ pack */

typedef struct {
  struct SLAEncoder* e;
  const int32_t* res;          /* final residual planes on host, stride = num_samples */
  const int32_t* const* pcm;   /* host PCM planes (for RAW blocks), may be NULL */
  uint8_t** bufs; uint32_t* sizes;
} pack_ctx_t;

static void pack_one(void* vctx, uint32_t b)
{
  pack_ctx_t* c = (pack_ctx_t*)vctx;
  struct SLAEncoder* e = c->e;
  const uint32_t C = e->wave_format.num_channels, O1 = e->encode_param.parcor_order + 1;
  const blk_t* k = &e->blk[b];
  slai_block_params bp;
  uint32_t rshift[SLAI_MAX_CHANNELS], pitch[SLAI_MAX_CHANNELS], rice[SLAI_MAX_CHANNELS], ch, s;
  int32_t ltm[SLAI_MAX_CHANNELS * SLAI_MAX_TAPS];
  int32_t* raw[SLAI_MAX_CHANNELS] = {0};
  uint32_t cap = 64 + C * (8 + 2 * O1 + 16);
  uint8_t* buf;
  memset(&bp, 0, sizeof(bp));
  bp.num_samples = k->nsmpl; bp.type = k->type; bp.num_channels = C; bp.order = O1 - 1;
  bp.ntaps = e->encode_param.longterm_order; bp.bps = e->wave_format.bit_per_sample; bp.lshift = e->lshift;
  bp.mid_side = (e->encode_param.ch_process_method == SLA_CHPROCESSMETHOD_STEREO_MS);
  for (ch = 0; ch < C; ch++) {
    const blkch_t* bc = &e->bc[(size_t)b * C + ch];
    rshift[ch] = bc->rshift; pitch[ch] = bc->pitch; rice[ch] = bc->rice_init;
    memcpy(&ltm[ch * SLAI_MAX_TAPS], bc->ltm_q, sizeof(int32_t) * SLAI_MAX_TAPS);
  }
  bp.code = e->code + (size_t)b * C * O1; bp.rshift = rshift; bp.pitch = pitch; bp.ltm_q = ltm; bp.rice_init = rice;
  if (k->type == SLAI_BLK_COMPRESS) {
    for (ch = 0; ch < C; ch++) { bp.res[ch] = c->res + (size_t)ch * e->num_samples + k->start; }
    cap += 8 * C * k->nsmpl;
  } else if (k->type == SLAI_BLK_RAW) {
    const uint32_t shift = 32 - bp.bps + e->lshift;
    for (ch = 0; ch < C; ch++) { raw[ch] = (int32_t*)malloc(sizeof(int32_t) * k->nsmpl); bp.res[ch] = raw[ch]; }
    for (s = 0; s < k->nsmpl; s++) {
      if (bp.mid_side) {
        const int32_t l = c->pcm[0][k->start + s] >> shift, r = c->pcm[1][k->start + s] >> shift;
        raw[0][s] = (int32_t)((uint32_t)l + (uint32_t)r) >> 1;
        raw[1][s] = (int32_t)((uint32_t)l - (uint32_t)r);
      } else {
        for (ch = 0; ch < C; ch++) { raw[ch][s] = c->pcm[ch][k->start + s] >> shift; }
      }
    }
    cap += 8 * C * k->nsmpl;
  }
  for (;;) {
    buf = (uint8_t*)malloc(cap);
    if (buf == NULL) { c->sizes[b] = 0; break; }
    c->sizes[b] = slai_pack_block(&bp, buf, cap);
    if (c->sizes[b] != 0 || cap > (1u << 30)) { break; }
    free(buf); buf = NULL;
    cap *= 4;                       /* pathological residuals: unary runs can be long */
  }
  c->bufs[b] = buf;
  for (ch = 0; ch < C; ch++) { free(raw[ch]); }
}

static int pack_impl(struct SLAEncoder* e, const int32_t* const* host_pcm, uint8_t* data, uint32_t data_size, uint32_t* output_size)
{
  const uint32_t C = e->wave_format.num_channels, n = e->num_samples;
  struct SLAHeaderInfo hdr;
  pack_ctx_t ctx;
  uint32_t b, ch, cur = SLA_HEADER_SIZE, maxblk = 0, maxbps = 0;
  const int32_t* planes[SLAI_MAX_CHANNELS];
  int need_raw = 0, rc = 0;
  if (!e->analysed) { return SLA_APIRESULT_PARAMETER_NOT_SET; }
  if (data == NULL || output_size == NULL) { return SLA_APIRESULT_INVALID_ARGUMENT; }
  if (data_size < SLA_HEADER_SIZE) { return SLA_APIRESULT_INSUFFICIENT_BUFFER_SIZE; }
  /* D2H of the final residual */
  RCCHK(pin_reserve(&e->h_res, sizeof(int32_t) * (size_t)C * (n + 1)));
  for (ch = 0; ch < C && n > 0 && RES2(e) != NULL; ch++) {
    HIPCHK(hipMemcpyAsync((int32_t*)e->h_res.ptr + (size_t)ch * n, RES2(e) + (size_t)ch * e->stride,
                          sizeof(int32_t) * n, hipMemcpyDeviceToHost, e->stream));
  }
  for (b = 0; b < e->num_blocks; b++) { if (e->blk[b].type == SLAI_BLK_RAW) { need_raw = 1; } }
  if (need_raw && host_pcm == NULL) {
    RCCHK(pin_reserve(&e->h_pcm, sizeof(int32_t) * (size_t)C * (n + 1)));
    for (ch = 0; ch < C; ch++) {
      HIPCHK(hipMemcpyAsync((int32_t*)e->h_pcm.ptr + (size_t)ch * n, e->pcm_dev + (size_t)ch * e->stride,
                            sizeof(int32_t) * n, hipMemcpyDeviceToHost, e->stream));
      planes[ch] = (const int32_t*)e->h_pcm.ptr + (size_t)ch * n;
    }
    host_pcm = planes;
  }
  HIPCHK(hipStreamSynchronize(e->stream));

  ctx.e = e; ctx.res = (const int32_t*)e->h_res.ptr; ctx.pcm = host_pcm;
  ctx.bufs = (uint8_t**)calloc(e->num_blocks + 1, sizeof(uint8_t*));
  ctx.sizes = (uint32_t*)calloc(e->num_blocks + 1, sizeof(uint32_t));
  if (ctx.bufs == NULL || ctx.sizes == NULL) { free(ctx.bufs); free(ctx.sizes); return SLA_APIRESULT_NG; }
  parallel_for(e->pool, e->num_blocks, pack_one, &ctx);

  for (b = 0; b < e->num_blocks; b++) {
    uint32_t bps_blk;
    if (ctx.bufs[b] == NULL || ctx.sizes[b] == 0) { rc = SLA_APIRESULT_NG; break; }
    if (cur >= data_size || ctx.sizes[b] > data_size - cur) { rc = SLA_APIRESULT_INSUFFICIENT_BUFFER_SIZE; break; }
    memcpy(data + cur, ctx.bufs[b], ctx.sizes[b]);
    e->blk[b].bytes = ctx.sizes[b];
    cur += ctx.sizes[b];
    if (ctx.sizes[b] > maxblk) { maxblk = ctx.sizes[b]; }
    bps_blk = (8 * ctx.sizes[b] * e->wave_format.sampling_rate) / e->blk[b].nsmpl;     /* src/SLAEncoder.c:895 */
    if (bps_blk > maxbps) { maxbps = bps_blk; }
  }
  for (b = 0; b < e->num_blocks; b++) { free(ctx.bufs[b]); }
  free(ctx.bufs); free(ctx.sizes);
  if (rc != 0) { return rc; }
  hdr.wave_format = e->wave_format; hdr.wave_format.offset_lshift = (uint8_t)e->lshift;
  hdr.encode_param = e->encode_param; hdr.num_samples = n; hdr.num_blocks = e->num_blocks;
  hdr.max_block_size = maxblk; hdr.max_bit_per_second = maxbps;
  rc = slai_write_header(&hdr, data, data_size);
  *output_size = cur;
  return rc;
}

int sla_hip_pack(struct SLAEncoder* e, uint8_t* data, uint32_t data_size, uint32_t* output_size)
{
  if (e == NULL) { return SLA_APIRESULT_INVALID_ARGUMENT; }
  RCCHK(enter(e));
  return pack_impl(e, NULL, data, data_size, output_size);
}

/* ------------------------------------------------------------------ device pack
 * Rice/Golomb/gamma coding, block assembly and CRC16 on the device.  The host only
 * packs the few header bytes of every block (it owns the per-block parameters), turns the per-channel
 * bit counts into block sizes/offsets, and writes the 43-byte file header. */
static int download_bytes(struct SLAEncoder* e, uint8_t* dst, const uint8_t* d_src, size_t bytes);
static int download_bytes_after(struct SLAEncoder* e, uint8_t* dst, const uint8_t* d_src, size_t bytes, hipEvent_t after);

/* One output file of a pack pass: the blocks that start inside [lo, hi) of the planes. */
typedef struct {
  uint32_t lo, hi;                       /* in  */
  uint8_t* data; uint32_t data_size;     /* in: the caller's buffer */
  uint32_t out_size; int result;         /* out */
  uint32_t num_blocks, max_block, max_bps;
  uint64_t img_off;                      /* where the file starts in the device image */
  /* a piece of a streamed file (one segment only): no 43-byte header in front of the blocks, and the destination is
   * asked for once the size is known -- place() sets data / data_size (the piece's place in the caller's buffer) */
  int bare;
  int (*place)(void* ctx, uint32_t out_size, uint8_t** data, uint32_t* data_size);
  void* place_ctx;
} pack_seg_t;

static int download_bytes(struct SLAEncoder* e, uint8_t* dst, const uint8_t* d_src, size_t bytes);

typedef struct {
  struct SLAEncoder* e; sla_hip_pack_block* pb; sla_hip_rice_job* jobs; uint8_t* hdr; const uint32_t* job_of; int bad;
  int what;                      /* 1: block records + Rice jobs (what the code-length kernels need), 2: header bytes */
} pack_hdr_ctx_t;

/* header bytes and Rice jobs of one block, at the places pack_device_core laid out */
static void pack_hdr_one(void* vctx, uint32_t b)
{
  pack_hdr_ctx_t* c = (pack_hdr_ctx_t*)vctx;
  struct SLAEncoder* e = c->e;
  const uint32_t C = e->wave_format.num_channels, O1 = e->encode_param.parcor_order + 1;
  const blk_t* k = &e->blk[b];
  sla_hip_pack_block* pb = &c->pb[b];
  slai_block_params bp;
  uint32_t rshift[SLAI_MAX_CHANNELS], pitch[SLAI_MAX_CHANNELS], rice[SLAI_MAX_CHANNELS], ch;
  int32_t ltm[SLAI_MAX_CHANNELS * SLAI_MAX_TAPS];
  uint8_t tmp[16];
  memset(&bp, 0, sizeof(bp));
  bp.num_samples = k->nsmpl; bp.type = k->type; bp.num_channels = C; bp.order = O1 - 1;
  bp.ntaps = e->encode_param.longterm_order; bp.bps = e->wave_format.bit_per_sample; bp.lshift = e->lshift;
  bp.mid_side = (e->encode_param.ch_process_method == SLA_CHPROCESSMETHOD_STEREO_MS);
  for (ch = 0; ch < C; ch++) {
    const blkch_t* bc = &e->bc[(size_t)b * C + ch];
    rshift[ch] = bc->rshift; pitch[ch] = bc->pitch; rice[ch] = bc->rice_init;
    memcpy(&ltm[ch * SLAI_MAX_TAPS], bc->ltm_q, sizeof(int32_t) * SLAI_MAX_TAPS);
  }
  bp.code = e->code + (size_t)b * C * O1; bp.rshift = rshift; bp.pitch = pitch; bp.ltm_q = ltm; bp.rice_init = rice;
  if (c->what == 1) {
    pb->blk_off = k->start; pb->num_samples = k->nsmpl; pb->type = k->type;
    pb->raw_bits = bp.bps - e->lshift;
    if (k->type == SLAI_BLK_COMPRESS) {
      slai_coding_mode(rice, C, pb->golomb_m);
      for (ch = 0; ch < C; ch++) {
        sla_hip_rice_job* j = &c->jobs[c->job_of[b] + ch];
        j->blk_off = k->start; j->blk_len = k->nsmpl; j->channel = ch; j->rice_init = rice[ch]; j->golomb_m = pb->golomb_m[ch];
      }
    }
    return;
  }
  if (pb->header_bytes >= 16) {
    if (slai_pack_header(&bp, c->hdr + pb->header_off, pb->header_bytes) != pb->header_bytes) { c->bad = 1; return; }
  } else {                                      /* (the packer wants 16 bytes of room: a silent block's header has 11) */
    const uint32_t got = slai_pack_header(&bp, tmp, sizeof(tmp));
    if (got != pb->header_bytes) { c->bad = 1; return; }
    memcpy(c->hdr + pb->header_off, tmp, got);
  }
}

/* Rice code lengths, block assembly and CRC16 on the device for every file of the analysed planes (segs sorted by
 * position, blocks are), one image holding the files back to back; each file is then copied to its buffer and gets
 * its 43-byte header.  A file that does not fit its buffer is reported in its `result`, the others are delivered. */
static int pack_device_core(struct SLAEncoder* e, pack_seg_t* segs, uint32_t nsegs)
{
  uint32_t C, O1, nb, b, ch, njobs = 0, sg;
  sla_hip_rice_job* jobs; sla_hip_pack_block* pb; uint8_t* hdr; uint32_t* job_of;
  size_t hdr_used = 0, hdr_cap;
  uint64_t cur = 0;
  struct SLAHeaderInfo hinfo;
  int rc = 0;
  C = e->wave_format.num_channels; O1 = e->encode_param.parcor_order + 1; nb = e->num_blocks;
  hdr_cap = (size_t)nb * (16 + C * (8 + 2 * O1 + 16)) + 64;
  RCCHK(pin_reserve(&e->h_pk_jobs, sizeof(sla_hip_rice_job) * ((size_t)nb * C + 1)));
  RCCHK(pin_reserve(&e->h_pk_blocks, sizeof(sla_hip_pack_block) * ((size_t)nb + 1)));
  RCCHK(pin_reserve(&e->h_pk_hdr, hdr_cap));
  RCCHK(pin_reserve(&e->h_fold, sizeof(uint64_t) * ((size_t)nb * C + 1)));
  jobs = (sla_hip_rice_job*)e->h_pk_jobs.ptr; pb = (sla_hip_pack_block*)e->h_pk_blocks.ptr; hdr = (uint8_t*)e->h_pk_hdr.ptr;
  job_of = (uint32_t*)malloc(sizeof(uint32_t) * ((size_t)nb + 1));
  if (job_of == NULL) { return SLA_APIRESULT_NG; }

  const double tp0 = now_ms();
  double tp1 = 0, tp2 = 0, tp3 = 0, tp4 = 0;
  /* per block: header bytes, coding mode, Rice jobs.  Sizes and places first (arithmetic only), then the host threads
   * write the headers and job records side by side */
  {
    const uint32_t order = O1 - 1, ntaps = e->encode_param.longterm_order, bps = e->wave_format.bit_per_sample;
    const uint32_t coef_bits = 16 * (order < 3 ? order : 3) + 8 * (order > 3 ? order - 3 : 0);
    pack_hdr_ctx_t hc;
    for (b = 0; b < nb; b++) {
      const blk_t* k = &e->blk[b];
      uint32_t bits = 16 + 32 + 16 + 16 + 2;                    /* src/SLAEncoder.c:682-741 */
      memset(&pb[b], 0, sizeof(pb[b]));
      if (k->type == SLAI_BLK_COMPRESS) {
        for (ch = 0; ch < C; ch++) {
          bits += 4 + coef_bits + 1 + bps;
          if (e->bc[(size_t)b * C + ch].pitch >= SLAI_LTM_MIN_PITCH) { bits += SLAI_LTM_PERIOD_BITS + 16 * ntaps; }
        }
      }
      pb[b].header_off = (uint32_t)hdr_used;
      pb[b].header_bytes = (bits + 7) / 8;
      hdr_used += pb[b].header_bytes;
      job_of[b] = njobs;
      if (k->type == SLAI_BLK_COMPRESS) { njobs += C; }
    }
    if (hdr_used > hdr_cap) { free(job_of); return SLA_APIRESULT_NG; }
    hc.e = e; hc.pb = pb; hc.jobs = jobs; hc.hdr = hdr; hc.job_of = job_of; hc.bad = 0; hc.what = 1;
    parallel_for(e->pool, nb, pack_hdr_one, &hc);
  }

  tp1 = now_ms();
  /* code lengths on the device */
  if (njobs > 0) {
    /* (every exit of this function releases job_of) */
    if ((rc = dev_reserve(&e->d_kk, sizeof(uint16_t) * (size_t)C * e->stride)) != 0
        || (rc = dev_reserve(&e->d_pk_jobs, sizeof(sla_hip_rice_job) * njobs)) != 0
        || (rc = dev_reserve(&e->d_fold, sizeof(uint64_t) * njobs)) != 0
        || (rc = hiprc(hipMemcpyAsync(e->d_pk_jobs.ptr, jobs, sizeof(sla_hip_rice_job) * njobs, hipMemcpyHostToDevice, e->stream))) != 0
        || (rc = sla_hip_launch_rice_len(RES2(e), e->stride, (const sla_hip_rice_job*)e->d_pk_jobs.ptr, njobs,
                                         (uint16_t*)e->d_kk.ptr, (uint64_t*)e->d_fold.ptr, e->stream)) != 0
        || (rc = hiprc(hipMemcpyAsync(e->h_fold.ptr, e->d_fold.ptr, sizeof(uint64_t) * njobs, hipMemcpyDeviceToHost, e->stream))) != 0) {
      free(job_of);
      return rc;
    }
  }
  {
    /* the header bytes, while the device walks the Rice parameters */
    pack_hdr_ctx_t hc;
    hc.e = e; hc.pb = pb; hc.jobs = jobs; hc.hdr = hdr; hc.job_of = job_of; hc.bad = 0; hc.what = 2;
    parallel_for(e->pool, nb, pack_hdr_one, &hc);
    if (hc.bad) { free(job_of); (void)hipStreamSynchronize(e->stream); return SLA_APIRESULT_NG; }
  }
  if (njobs > 0 && (rc = hiprc(hipStreamSynchronize(e->stream))) != 0) { free(job_of); return rc; }

  tp2 = now_ms();
  /* block sizes -> offsets; a file = 43 header bytes + its blocks */
  for (sg = 0; sg < nsegs; sg++) { segs[sg].out_size = 0; segs[sg].result = 0; segs[sg].num_blocks = 0; segs[sg].max_block = 0; segs[sg].max_bps = 0; }
  sg = 0; segs[0].img_off = 0; cur = (nsegs == 1 && segs[0].bare) ? 0 : SLA_HEADER_SIZE;
  for (b = 0; b < nb; b++) {
    const blk_t* k = &e->blk[b];
    uint64_t body_bits = 0, bytes;
    uint32_t bps_blk;
    while (sg + 1 < nsegs && k->start >= segs[sg].hi) {
      segs[sg].out_size = (uint32_t)(cur - segs[sg].img_off);
      sg++;
      segs[sg].img_off = cur; cur += SLA_HEADER_SIZE;
    }
    if (k->type == SLAI_BLK_COMPRESS) {
      for (ch = 0; ch < C; ch++) { body_bits += ((const uint64_t*)e->h_fold.ptr)[job_of[b] + ch]; }
    } else if (k->type == SLAI_BLK_RAW) {
      uint64_t per_sample = (uint64_t)C * pb[b].raw_bits
        + ((e->encode_param.ch_process_method == SLA_CHPROCESSMETHOD_STEREO_MS) ? 1u : 0u);
      body_bits = per_sample * k->nsmpl;
    }
    bytes = pb[b].header_bytes + (body_bits + 7) / 8;
    if (bytes > 0xFFFFFFF0ull) { free(job_of); return SLA_APIRESULT_INSUFFICIENT_BUFFER_SIZE; }
    if (cur - segs[sg].img_off >= segs[sg].data_size || bytes > (uint64_t)segs[sg].data_size - (cur - segs[sg].img_off)) {
      segs[sg].result = SLA_APIRESULT_INSUFFICIENT_BUFFER_SIZE;
    }
    pb[b].out_off = cur; pb[b].out_bytes = (uint32_t)bytes;
    e->blk[b].bytes = (uint32_t)bytes;
    cur += bytes;
    segs[sg].num_blocks++;
    if (bytes > segs[sg].max_block) { segs[sg].max_block = (uint32_t)bytes; }
    bps_blk = (8 * (uint32_t)bytes * e->wave_format.sampling_rate) / k->nsmpl;        /* src/SLAEncoder.c:895 */
    if (bps_blk > segs[sg].max_bps) { segs[sg].max_bps = bps_blk; }
  }
  for (;;) {                               /* close the file of the last block and the block-less files behind it */
    segs[sg].out_size = (uint32_t)(cur - segs[sg].img_off);
    if (sg + 1 >= nsegs) { break; }
    sg++;
    segs[sg].img_off = cur; cur += SLA_HEADER_SIZE;
  }
  free(job_of);
  if (nsegs == 1 && segs[0].place != NULL) {
    rc = segs[0].place(segs[0].place_ctx, segs[0].out_size, &segs[0].data, &segs[0].data_size);
    if (rc != 0) { return rc; }
    if (segs[0].out_size > segs[0].data_size) { segs[0].result = SLA_APIRESULT_INSUFFICIENT_BUFFER_SIZE; }
  }
  if (nsegs == 1 && segs[0].result != 0) { return segs[0].result; }     /* one file: nothing worth assembling */

  /* assemble the image on the device, bring it back with one copy */
  {
    const size_t img_bytes = ((size_t)cur + 8 + 3) & ~(size_t)3;
    RCCHK(dev_reserve(&e->d_image, img_bytes));
    RCCHK(dev_reserve(&e->d_pk_blocks, sizeof(sla_hip_pack_block) * (nb + 1)));
    RCCHK(dev_reserve(&e->d_pk_hdr, hdr_used + 16));
    if (e->d_kk.ptr == NULL) { RCCHK(dev_reserve(&e->d_kk, sizeof(uint16_t) * (size_t)C * e->stride)); }
    HIPCHK(hipMemsetAsync(e->d_image.ptr, 0, img_bytes, e->stream));
    /* one file of some size: the blocks are written in up to four runs of about equal bytes, and a run's bytes leave
     * on the download stream while the next run is written (blocks are byte-aligned and only ever OR into their own
     * bytes, so a finished run is final even where it shares a word with its neighbour) */
    uint32_t nruns = (nsegs == 1 && nb >= 64 && cur >= (8u << 20)) ? 4u : 1u, run_lo[5], r;
    run_lo[0] = 0;
    for (r = 1; r < nruns; r++) {
      uint32_t lo = run_lo[r - 1], hi = nb;
      const uint64_t want = segs[0].img_off + (cur - segs[0].img_off) * r / nruns;
      while (lo < hi) { const uint32_t m = (lo + hi) / 2; if (pb[m].out_off < want) { lo = m + 1; } else { hi = m; } }
      run_lo[r] = lo;
    }
    run_lo[nruns] = nb;
    if (nb > 0) {
      HIPCHK(hipMemcpyAsync(e->d_pk_blocks.ptr, pb, sizeof(sla_hip_pack_block) * nb, hipMemcpyHostToDevice, e->stream));
      HIPCHK(hipMemcpyAsync(e->d_pk_hdr.ptr, hdr, hdr_used, hipMemcpyHostToDevice, e->stream));
      for (r = 0; r < nruns; r++) {
        if (run_lo[r + 1] > run_lo[r]) {
          RCCHK(sla_hip_launch_rice_write(RES2(e), e->pcm_dev, e->stride, (const uint16_t*)e->d_kk.ptr,
                                          (const sla_hip_pack_block*)e->d_pk_blocks.ptr + run_lo[r], run_lo[r + 1] - run_lo[r],
                                          (const uint8_t*)e->d_pk_hdr.ptr, C, 32 - e->wave_format.bit_per_sample + e->lshift,
                                          e->encode_param.ch_process_method == SLA_CHPROCESSMETHOD_STEREO_MS,
                                          (uint32_t*)e->d_image.ptr, e->stream));
        }
        if (nruns > 1) { HIPCHK(hipEventRecord(e->ev_pack[r], e->stream)); }
      }
    }
    tp3 = now_ms();
    if (nruns > 1 && segs[0].result == 0) {
      for (r = 0; r < nruns; r++) {
        const uint64_t from = (r == 0) ? segs[0].img_off : pb[run_lo[r]].out_off;
        const uint64_t upto = (r + 1 == nruns || run_lo[r + 1] >= nb) ? cur : pb[run_lo[r + 1]].out_off;
        if (r > 0 && run_lo[r] >= nb) { break; }
        RCCHK(download_bytes_after(e, segs[0].data + (from - segs[0].img_off), (const uint8_t*)e->d_image.ptr + from, (size_t)(upto - from), e->ev_pack[r]));
      }
    } else {
      for (sg = 0; sg < nsegs; sg++) {
        if (segs[sg].result != 0) { continue; }
        RCCHK(download_bytes(e, segs[sg].data, (const uint8_t*)e->d_image.ptr + segs[sg].img_off, (size_t)segs[sg].out_size));
      }
    }
    tp4 = now_ms();
  }
  if (e->trace) {
    fprintf(stderr, "[sla_hip] pack: headers + jobs %.3f ms, k_rice_len + sizes home %.3f ms, offsets + launches %.3f ms, write/crc kernels + download %.3f ms\n",
            tp1 - tp0, tp2 - tp1, tp3 - tp2, tp4 - tp3);
  }
  for (sg = 0; sg < nsegs; sg++) {
    if (segs[sg].result != 0 || segs[sg].bare) { continue; }
    hinfo.wave_format = e->wave_format; hinfo.wave_format.offset_lshift = (uint8_t)e->lshift;
    hinfo.encode_param = e->encode_param; hinfo.num_samples = segs[sg].hi - segs[sg].lo; hinfo.num_blocks = segs[sg].num_blocks;
    hinfo.max_block_size = segs[sg].max_block; hinfo.max_bit_per_second = segs[sg].max_bps;
    rc = slai_write_header(&hinfo, segs[sg].data, segs[sg].data_size);
    if (rc != 0) { segs[sg].result = rc; }
  }
  return 0;
}

int sla_hip_pack_device(struct SLAEncoder* e, uint8_t* data, uint32_t data_size, uint32_t* output_size)
{
  pack_seg_t seg;
  int rc;
  if (e == NULL) { return SLA_APIRESULT_INVALID_ARGUMENT; }
  if (!e->analysed) { return SLA_APIRESULT_PARAMETER_NOT_SET; }
  if (data == NULL || output_size == NULL) { return SLA_APIRESULT_INVALID_ARGUMENT; }
  if (data_size < SLA_HEADER_SIZE) { return SLA_APIRESULT_INSUFFICIENT_BUFFER_SIZE; }
  RCCHK(enter(e));
  memset(&seg, 0, sizeof(seg));
  seg.lo = 0; seg.hi = e->num_samples; seg.data = data; seg.data_size = data_size;
  rc = pack_device_core(e, &seg, 1);
  if (rc != 0) { return rc; }
  if (seg.result != 0) { return seg.result; }
  *output_size = seg.out_size;
  return 0;
}

/* -------------------------------------------------------------- public encode API */

/* ---- PCIe path: pageable host memory <-> device through two pinned staging slots.  Host threads fill
 * (or drain) one slot while the other one is on the bus; input of <= 16 significant bits crosses as int16. */
#define XFER_SLOT_BYTES (8u << 20)
#define XFER_GRAIN      (64u << 10)          /* samples / bytes per host-thread work item */

typedef struct { const int32_t* src; int16_t* dst16; uint8_t* dst24; int32_t* dst32; size_t count; uint32_t lowbits; } stage_in_t;
static void stage_in_one(void* vctx, uint32_t i)
{
  stage_in_t* c = (stage_in_t*)vctx;
  const size_t lo = (size_t)i * XFER_GRAIN, hi = (lo + XFER_GRAIN < c->count) ? lo + XFER_GRAIN : c->count;
  size_t k;
  if (c->dst16 != NULL) {
    uint32_t low = 0;
    for (k = lo; k < hi; k++) { const int32_t v = c->src[k]; low |= (uint32_t)v & 0xFFFFu; c->dst16[k] = (int16_t)(v >> 16); }
    if (low) { __atomic_fetch_or(&c->lowbits, low, __ATOMIC_RELAXED); }
  } else if (c->dst24 != NULL) {
    /* option "upload24": the three significant bytes of every sample, four samples = three 32-bit words (XFER_GRAIN is a
     * multiple of 4: every work item starts on a word) */
    uint32_t low = 0;
    uint32_t* w = (uint32_t*)(c->dst24 + 3 * lo);
    for (k = lo; k + 4 <= hi; k += 4, w += 3) {
      const uint32_t a = (uint32_t)c->src[k], b = (uint32_t)c->src[k + 1], d = (uint32_t)c->src[k + 2], f = (uint32_t)c->src[k + 3];
      low |= (a | b | d | f) & 0xFFu;
      w[0] = (a >> 8) | ((b >> 8) << 24);
      w[1] = (b >> 16) | ((d >> 8) << 16);
      w[2] = (d >> 24) | ((f >> 8) << 8);
    }
    for (; k < hi; k++) {
      const uint32_t a = (uint32_t)c->src[k];
      low |= a & 0xFFu;
      c->dst24[3 * k] = (uint8_t)(a >> 8); c->dst24[3 * k + 1] = (uint8_t)(a >> 16); c->dst24[3 * k + 2] = (uint8_t)(a >> 24);
    }
    if (low) { __atomic_fetch_or(&c->lowbits, low, __ATOMIC_RELAXED); }
  } else {
    memcpy(c->dst32 + lo, c->src + lo, sizeof(int32_t) * (hi - lo));
  }
}

/* mode16: 0 = int32 as it is, 1 = int16 (the upper half), 24 = three bytes per sample (option "upload24") */
static int upload_pass(struct SLAEncoder* e, const int32_t* const* input, uint32_t n, uint64_t stride, int mode16, uint32_t* lowbits)
{
  extern int sla_hip_launch_unpack16(const int16_t*, int32_t*, uint64_t, sla_hip_stream_t);
  extern int sla_hip_launch_unpack24(const uint8_t*, int32_t*, uint64_t, sla_hip_stream_t);
  const uint32_t C = e->wave_format.num_channels;
  const int mode24 = (mode16 == 24);
  const size_t slot_samples = mode24 ? ((XFER_SLOT_BYTES / 3) & ~(size_t)(XFER_GRAIN - 1)) : mode16 ? (XFER_SLOT_BYTES / 2) : (XFER_SLOT_BYTES / 4);
  uint32_t ch, k = 0;
  size_t o;
  stage_in_t ctx;
  ctx.lowbits = 0;
  for (ch = 0; ch < C; ch++) {
    for (o = 0; o < n; o += slot_samples, k++) {
      const uint32_t slot = k & 1;
      const size_t count = (n - o < slot_samples) ? (n - o) : slot_samples;
      int32_t* dst = (int32_t*)e->d_pcm.ptr + (size_t)ch * stride + o;
      if (k >= 2) { HIPCHK(hipEventSynchronize(e->ev_stage[slot])); }
      ctx.src = input[ch] + o; ctx.count = count;
      ctx.dst16 = (mode16 && !mode24) ? (int16_t*)e->h_stage[slot].ptr : NULL;
      ctx.dst24 = mode24 ? (uint8_t*)e->h_stage[slot].ptr : NULL;
      ctx.dst32 = mode16 ? NULL : (int32_t*)e->h_stage[slot].ptr;
      parallel_for(e->upload_pool != NULL ? e->upload_pool : e->pool, (uint32_t)((count + XFER_GRAIN - 1) / XFER_GRAIN), stage_in_one, &ctx);
      if (mode24) {
        HIPCHK(hipMemcpyAsync(e->d_stage[slot].ptr, e->h_stage[slot].ptr, count * 3, hipMemcpyHostToDevice, e->stream));
        RCCHK(sla_hip_launch_unpack24((const uint8_t*)e->d_stage[slot].ptr, dst, count, e->stream));
      } else if (mode16) {
        HIPCHK(hipMemcpyAsync(e->d_stage[slot].ptr, e->h_stage[slot].ptr, count * 2, hipMemcpyHostToDevice, e->stream));
        RCCHK(sla_hip_launch_unpack16((const int16_t*)e->d_stage[slot].ptr, dst, count, e->stream));
      } else {
        HIPCHK(hipMemcpyAsync(dst, e->h_stage[slot].ptr, count * 4, hipMemcpyHostToDevice, e->stream));
      }
      HIPCHK(hipEventRecord(e->ev_stage[slot], e->stream));
    }
  }
  HIPCHK(hipStreamSynchronize(e->stream));
  *lowbits = ctx.lowbits;
  return 0;
}

/* 1 when [p, p + bytes) is page-locked host memory the runtime knows (hipHostMalloc / hipHostRegister): such a
 * buffer crosses the bus by DMA as it is, without the copy through the pinned staging slots */
static int is_pinned_host(const void* p, size_t bytes)
{
  hipPointerAttribute_t at;
  if (bytes == 0) { return 0; }
  memset(&at, 0, sizeof(at));
  if (hipPointerGetAttributes(&at, p) != hipSuccess) { (void)hipGetLastError(); return 0; }
  if (at.type != hipMemoryTypeHost) { return 0; }
  memset(&at, 0, sizeof(at));
  if (hipPointerGetAttributes(&at, (const uint8_t*)p + bytes - 1) != hipSuccess) { (void)hipGetLastError(); return 0; }
  return at.type == hipMemoryTypeHost;
}

static int upload_pcm(struct SLAEncoder* e, const int32_t* const* input, uint32_t n)
{
  const uint32_t C = e->wave_format.num_channels;
  const uint64_t stride = ((uint64_t)n + 63) & ~(uint64_t)63;
  uint32_t ch, lowbits = 0;
  int mode16 = (e->wave_format.bit_per_sample <= 16) ? 1 : (e->upload24 && e->wave_format.bit_per_sample <= 24) ? 24 : 0, pinned = (n > 0);
  for (ch = 0; ch < C; ch++) { if (input[ch] == NULL) { return SLA_APIRESULT_INVALID_ARGUMENT; } }
  RCCHK(dev_reserve(&e->d_pcm, sizeof(int32_t) * (size_t)C * (stride + 64)));
  for (ch = 0; ch < C && pinned; ch++) { pinned = is_pinned_host(input[ch], sizeof(int32_t) * (size_t)n); }
  if (pinned) {
    /* the caller's planes are page-locked: one DMA per channel at the bus rate, nothing for the host threads to do */
    for (ch = 0; ch < C; ch++) {
      size_t o;
      for (o = 0; o < n; o += XFER_SLOT_BYTES / 4) {       /* (pieces of the staging slots' size: measured faster than one copy per plane) */
        const size_t cnt = (n - o < XFER_SLOT_BYTES / 4) ? (n - o) : XFER_SLOT_BYTES / 4;
        HIPCHK(hipMemcpyAsync((int32_t*)e->d_pcm.ptr + (size_t)ch * stride + o, input[ch] + o, sizeof(int32_t) * cnt, hipMemcpyHostToDevice, e->stream));
      }
    }
    HIPCHK(hipStreamSynchronize(e->stream));
    e->pcm_dev = (const int32_t*)e->d_pcm.ptr; e->stride = stride; e->num_samples = n;
    return 0;
  }
  for (ch = 0; ch < 2; ch++) {
    RCCHK(pin_reserve(&e->h_stage[ch], XFER_SLOT_BYTES));
    RCCHK(dev_reserve(&e->d_stage[ch], XFER_SLOT_BYTES));
  }
  RCCHK(upload_pass(e, input, n, stride, mode16, &lowbits));
  if (mode16 && lowbits != 0) {            /* not really <= 16-bit data: keep every bit, as the reference would */
    RCCHK(upload_pass(e, input, n, stride, 0, &lowbits));
  }
  e->pcm_dev = (const int32_t*)e->d_pcm.ptr; e->stride = stride; e->num_samples = n;
  return 0;
}

typedef struct { const uint8_t* src; uint8_t* dst; size_t count; } stage_out_t;
static void stage_out_one(void* vctx, uint32_t i)
{
  stage_out_t* c = (stage_out_t*)vctx;
  const size_t lo = (size_t)i * XFER_GRAIN, hi = (lo + XFER_GRAIN < c->count) ? lo + XFER_GRAIN : c->count;
  memcpy(c->dst + lo, c->src + lo, hi - lo);
}

/* device -> host: page-locked destinations by DMA, pageable ones double-buffered through the pinned slots.  `after` (may
 * be NULL): the copies run on the download stream once that event has happened, beside later kernels of the handle's
 * main stream; without it they follow whatever the main stream holds */
static int download_bytes_after(struct SLAEncoder* e, uint8_t* dst, const uint8_t* d_src, size_t bytes, hipEvent_t after)
{
  size_t o, k = 0, nslots = (bytes + XFER_SLOT_BYTES - 1) / XFER_SLOT_BYTES;
  hipStream_t st = (after != NULL) ? e->stream_down : e->stream;
  stage_out_t ctx;
  uint32_t s;
  if (bytes == 0) { return 0; }
  if (after != NULL) { HIPCHK(hipStreamWaitEvent(st, after, 0)); }
  if (is_pinned_host(dst, bytes)) {            /* page-locked destination: straight DMA */
    for (o = 0; o < bytes; o += XFER_SLOT_BYTES) {
      HIPCHK(hipMemcpyAsync(dst + o, d_src + o, (bytes - o < XFER_SLOT_BYTES) ? (bytes - o) : XFER_SLOT_BYTES, hipMemcpyDeviceToHost, st));
    }
    HIPCHK(hipStreamSynchronize(st));
    return 0;
  }
  for (s = 0; s < 2; s++) { RCCHK(pin_reserve(&e->h_stage[s], XFER_SLOT_BYTES)); }
  for (k = 0; k <= nslots; k++) {
    if (k < nslots) {
      o = k * XFER_SLOT_BYTES;
      HIPCHK(hipMemcpyAsync(e->h_stage[k & 1].ptr, d_src + o, (bytes - o < XFER_SLOT_BYTES) ? (bytes - o) : XFER_SLOT_BYTES,
                            hipMemcpyDeviceToHost, st));
      HIPCHK(hipEventRecord(e->ev_stage[k & 1], st));
    }
    if (k >= 1) {
      o = (k - 1) * XFER_SLOT_BYTES;
      HIPCHK(hipEventSynchronize(e->ev_stage[(k - 1) & 1]));
      ctx.src = (const uint8_t*)e->h_stage[(k - 1) & 1].ptr; ctx.dst = dst + o;
      ctx.count = (bytes - o < XFER_SLOT_BYTES) ? (bytes - o) : XFER_SLOT_BYTES;
      parallel_for(e->pool, (uint32_t)((ctx.count + XFER_GRAIN - 1) / XFER_GRAIN), stage_out_one, &ctx);
    }
  }
  return 0;
}

static int download_bytes(struct SLAEncoder* e, uint8_t* dst, const uint8_t* d_src, size_t bytes)
{
  return download_bytes_after(e, dst, d_src, bytes, NULL);
}

/* ---- SLAEncoder_EncodeWhole of a long file, streamed ------------------------------------------------------------
 * One file = upload (PCIe in), analysis (kernels), pack (kernels), download (PCIe out): four resources that the plain
 * path uses one after the other.  Blocks are independent and the only sequential thing in the format is
 * the super-frame hop, so the file is cut into pieces exactly as "one file, several GPUs" cuts it (sla_hip.h) -- only
 * that the "ranks" are worker lanes on ONE device: handles of their own (own streams, buffers, host threads); a lane that
 * has delivered its piece takes the next piece nobody has taken.  A lane uploads its piece (+ one block of halo), scans it, takes over the hop where the piece
 * before it ended, analyses [bounds[r], bounds[r+1]), sizes its blocks, learns where they go in the caller's buffer
 * from the piece before it, and writes them there.  Uploads take turns in piece order (piece 0 should be analysed
 * while piece 1 is still on the bus, not share the bus with it).
 * offset_lshift is a property of the whole file (trailing zeros of the OR of every sample, src/SLAEncoder.c:425-455):
 * the lanes take it from piece 0 and every later piece checks its own samples against it; a piece that disagrees
 * (or an all-zero piece 0) abandons the streamed pass and the file takes the plain path -- same bytes either way. */
typedef struct {
  struct SLAEncoder* parent;
  const int32_t* const* input; uint32_t n;
  uint8_t* data; uint32_t data_size;
  uint32_t K, L, maxb;
  uint32_t nominal[65];              /* piece r scans [nominal[r], nominal[r+1]) (multiples of 64) */
  pthread_mutex_t mu; pthread_cond_t cv;
  uint32_t upload_turn;              /* the piece whose upload may start */
  uint32_t next_piece;               /* the next piece nobody has taken: a lane that has delivered its piece takes this one (round 3 dealt
                                        piece r to lane r mod L: an upload then waited for ITS lane while another stood idle) */
  uint8_t  lane_of[65];              /* (for the trace) */
  int      have_ntz; uint32_t ntz;   /* trailing zeros of piece 0's OR word */
  uint32_t bounds[65]; uint32_t bounds_known;      /* bounds[0..bounds_known) are final */
  uint64_t off[65]; uint32_t off_known;            /* byte offset of piece r's first block in the file */
  uint32_t num_blocks, max_block, max_bps, lshift;
  int      failed;                   /* > 0: API result to return; < 0: take the plain path */
  double   t0; double stamp[65][7];  /* SLA_HIP_TRACE: per piece -- upload begins / ends, scanned, hop known, analysed, sized, delivered */
} stream_ctx_t;

typedef struct { stream_ctx_t* sc; uint32_t lane; } stream_arg_t;

static void stream_fail(stream_ctx_t* sc, int code)
{
  pthread_mutex_lock(&sc->mu);
  if (sc->failed == 0) { sc->failed = code; }
  pthread_cond_broadcast(&sc->cv);
  pthread_mutex_unlock(&sc->mu);
}

typedef struct { stream_ctx_t* sc; uint32_t r; } place_arg_t;

/* pack callback: piece r knows its size -- wait for the place the piece before it ends at, publish our own end */
static int stream_place(void* vctx, uint32_t out_size, uint8_t** data, uint32_t* data_size)
{
  place_arg_t* pa = (place_arg_t*)vctx;
  stream_ctx_t* sc = pa->sc;
  uint64_t at;
  int rc = 0;
  pthread_mutex_lock(&sc->mu);
  while (sc->off_known <= pa->r && sc->failed == 0) { pthread_cond_wait(&sc->cv, &sc->mu); }
  if (sc->failed != 0) { pthread_mutex_unlock(&sc->mu); return SLA_APIRESULT_NG; }
  at = sc->off[pa->r];
  sc->stamp[pa->r][5] = now_ms() - sc->t0;
  sc->off[pa->r + 1] = at + out_size;
  sc->off_known = pa->r + 2;
  pthread_cond_broadcast(&sc->cv);
  pthread_mutex_unlock(&sc->mu);
  if (at + out_size > (uint64_t)sc->data_size) { rc = SLA_APIRESULT_INSUFFICIENT_BUFFER_SIZE; }
  else { *data = sc->data + at; *data_size = (uint32_t)(sc->data_size - at); }
  return rc;
}

static void* stream_lane_main(void* varg)
{
  stream_arg_t* sa = (stream_arg_t*)varg;
  stream_ctx_t* sc = sa->sc;
  struct SLAEncoder* e = sc->parent->lane[sa->lane];
  const uint32_t C = e->wave_format.num_channels;
  const int32_t* planes[SLAI_MAX_CHANNELS];
  uint64_t* mask = NULL; size_t mask_cap = 0;
  uint32_t r, ch;
  for (;;) {
    pthread_mutex_lock(&sc->mu);
    r = sc->next_piece++;
    if (r < sc->K) { sc->lane_of[r] = (uint8_t)sa->lane; }
    pthread_mutex_unlock(&sc->mu);
    if (r >= sc->K) { break; }
    const uint32_t lo = sc->nominal[r], nominal_hi = sc->nominal[r + 1];
    const uint32_t hi = (sc->n - nominal_hi > sc->maxb) ? nominal_hi + sc->maxb : sc->n;      /* + the halo the hop may run into */
    const uint32_t cnt = hi - lo;
    const size_t nwords = ((size_t)cnt + 63) / 64;
    uint32_t orw = 0, b_lo, b_hi, pos, word;
    int rc;
    /* upload, in piece order */
    pthread_mutex_lock(&sc->mu);
    while (sc->upload_turn != r && sc->failed == 0) { pthread_cond_wait(&sc->cv, &sc->mu); }
    pthread_mutex_unlock(&sc->mu);
    if (sc->failed != 0) { break; }
    sc->stamp[r][0] = now_ms() - sc->t0;
    for (ch = 0; ch < C; ch++) { planes[ch] = sc->input[ch] + lo; }
    e->upload_pool = sc->parent->pool;
    rc = (enter(e) != 0) ? SLA_APIRESULT_NG : upload_pcm(e, planes, cnt);
    e->upload_pool = NULL;
    pthread_mutex_lock(&sc->mu);
    sc->upload_turn = r + 1;
    pthread_cond_broadcast(&sc->cv);
    pthread_mutex_unlock(&sc->mu);
    sc->stamp[r][1] = now_ms() - sc->t0;
    if (rc != 0) { stream_fail(sc, rc > 0 ? rc : SLA_APIRESULT_NG); break; }
    /* scan: OR word and silence mask of the piece and its halo */
    if (nwords + 2 > mask_cap) {
      free(mask);
      mask_cap = nwords + 2;
      mask = (uint64_t*)malloc(mask_cap * 8);
      if (mask == NULL) { stream_fail(sc, SLA_APIRESULT_NG); break; }
    }
    rc = sla_hip_shard_scan(e, e->pcm_dev, e->stride, cnt, &orw, mask);
    if (rc != 0) { stream_fail(sc, rc > 0 ? rc : SLA_APIRESULT_NG); break; }
    mask[nwords] = 0; mask[nwords + 1] = 0;
    sc->stamp[r][2] = now_ms() - sc->t0;
    /* offset_lshift: piece 0 speaks for the file, everybody else checks */
    pthread_mutex_lock(&sc->mu);
    if (r == 0) {
      if (orw == 0) { sc->failed = -1; } else { sc->ntz = (uint32_t)__builtin_ctz(orw); sc->have_ntz = 1; }
      pthread_cond_broadcast(&sc->cv);
    }
    while (!sc->have_ntz && sc->failed == 0) { pthread_cond_wait(&sc->cv, &sc->mu); }
    if (sc->failed == 0 && orw != 0 && (uint32_t)__builtin_ctz(orw) < sc->ntz) { sc->failed = -1; pthread_cond_broadcast(&sc->cv); }
    /* the hop: from where the piece before us ended, through our piece */
    while (sc->bounds_known <= r && sc->failed == 0) { pthread_cond_wait(&sc->cv, &sc->mu); }
    if (sc->failed != 0) { pthread_mutex_unlock(&sc->mu); break; }
    b_lo = sc->bounds[r];
    pthread_mutex_unlock(&sc->mu);
    sc->stamp[r][3] = now_ms() - sc->t0;
    pos = b_lo;
    if (r + 1 == sc->K) {
      pos = sc->n;
    } else {
      while (pos < nominal_hi) {                                  /* src/SLAEncoder.c:846-869, 392-408 */
        const uint32_t remain = sc->n - pos;
        const uint32_t window = (sc->maxb < remain) ? sc->maxb : remain;
        const uint32_t min_blk = (SLAI_MIN_BLOCK < remain) ? SLAI_MIN_BLOCK : remain;
        const uint32_t run = slai_zero_run(mask, pos - lo, window);
        pos += (run >= min_blk) ? run : window;
      }
    }
    b_hi = pos;
    pthread_mutex_lock(&sc->mu);
    sc->bounds[r + 1] = b_hi;
    sc->bounds_known = r + 2;
    pthread_cond_broadcast(&sc->cv);
    pthread_mutex_unlock(&sc->mu);
    if (b_hi == b_lo) {                                           /* nothing starts in this piece */
      pthread_mutex_lock(&sc->mu);
      while (sc->off_known <= r && sc->failed == 0) { pthread_cond_wait(&sc->cv, &sc->mu); }
      if (sc->failed == 0) { sc->off[r + 1] = sc->off[r]; sc->off_known = r + 2; pthread_cond_broadcast(&sc->cv); }
      pthread_mutex_unlock(&sc->mu);
      continue;
    }
    /* the hot path on our range, with the file's sample unit */
    word = 0xFFFFFFFFu << sc->ntz;
    rc = sla_hip_shard_analyze(e, e->pcm_dev + (b_lo - lo), e->stride, b_hi - b_lo, word, NULL);
    if (rc != 0) { stream_fail(sc, rc > 0 ? rc : SLA_APIRESULT_NG); break; }
    sc->stamp[r][4] = now_ms() - sc->t0;
    {
      pack_seg_t seg;
      place_arg_t pa;
      memset(&seg, 0, sizeof(seg));
      pa.sc = sc; pa.r = r;
      seg.lo = 0; seg.hi = b_hi - b_lo; seg.data = NULL; seg.data_size = 0xFFFFFFFFu; seg.bare = 1;
      seg.place = stream_place; seg.place_ctx = &pa;
      rc = (enter(e) != 0) ? SLA_APIRESULT_NG : pack_device_core(e, &seg, 1);
      if (rc == 0 && seg.result != 0) { rc = seg.result; }
      if (rc != 0) { stream_fail(sc, rc > 0 ? rc : SLA_APIRESULT_NG); break; }
      pthread_mutex_lock(&sc->mu);
      sc->num_blocks += seg.num_blocks;
      if (seg.max_block > sc->max_block) { sc->max_block = seg.max_block; }
      if (seg.max_bps > sc->max_bps) { sc->max_bps = seg.max_bps; }
      sc->lshift = e->lshift;
      pthread_mutex_unlock(&sc->mu);
      sc->stamp[r][6] = now_ms() - sc->t0;
    }
  }
  free(mask);
  return NULL;
}

/* a lane: a handle like its parent, fewer host threads; knobs and formats are copied at every use */
static struct SLAEncoder* stream_lane(struct SLAEncoder* e, uint32_t t)
{
  struct SLAEncoder* l = e->lane[t];
  if (l == NULL) {
    l = SLAEncoder_Create(&e->cfg);
    if (l == NULL) { return NULL; }
    l->is_lane = 1; l->stream_mode = 0;
    if (l->threads > 4) { pool_destroy(l->pool); l->threads = 4; l->pool = pool_create(4); if (l->pool == NULL) { SLAEncoder_Destroy(l); return NULL; } }
    e->lane[t] = l;
  }
  l->wave_format = e->wave_format; l->encode_param = e->encode_param; l->status_flag = e->status_flag; l->analysed = 0;
  l->chunks = e->chunks; l->chunks_forced = e->chunks_forced; l->first_chunk = e->first_chunk; l->split_count = 0;
  l->fuse_lattice = e->fuse_lattice; l->device_plan = e->device_plan; l->search_exact = e->search_exact; l->exact_bits = e->exact_bits;
  l->cert_safety = e->cert_safety; l->single_tail = e->single_tail; l->device_ltm = e->device_ltm; l->tune = e->tune;
  l->block_cert = e->block_cert; l->block_cert_safety = e->block_cert_safety; l->alt_streams = e->alt_streams; l->device_expand = e->device_expand; l->expand_silence = e->expand_silence;
  l->table_cache = e->table_cache; l->prelaunch = e->prelaunch; l->one_stream = e->one_stream;
  l->trace = 0;
  return l;
}

/* 0: done; < 0: not streamed (too short, switched off, or the pieces disagreed on offset_lshift) -- take the plain path;
 * > 0: the API result to return */
static int encode_whole_streamed(struct SLAEncoder* e, const int32_t* const* input, uint32_t n, uint8_t* data, uint32_t data_size,
                                 uint32_t* output_size)
{
  const uint32_t C = e->wave_format.num_channels;
  stream_ctx_t* sc;
  stream_arg_t args[SLAI_STREAM_LANES];
  pthread_t th[SLAI_STREAM_LANES];
  uint32_t piece, K, L, r, t, started = 0;
  int rc = 0;
  struct SLAHeaderInfo hinfo;
  if (!e->stream_mode || e->is_lane || C == 0) { return -1; }
  piece = e->stream_piece / C;
  piece = (piece + 63u) & ~63u;
  if (piece < e->encode_param.max_num_block_samples * 2u) { piece = (e->encode_param.max_num_block_samples * 2u + 63u) & ~63u; }
  if ((uint64_t)n < 2ull * piece) { return -1; }
  K = (uint32_t)(((uint64_t)n + piece - 1) / piece);
  if (K > 64) { K = 64; piece = (uint32_t)((((uint64_t)n + K - 1) / K + 63u) & ~(uint64_t)63u); K = (uint32_t)(((uint64_t)n + piece - 1) / piece); }
  L = (e->stream_lanes < K) ? e->stream_lanes : K;
  for (t = 0; t < L; t++) { if (stream_lane(e, t) == NULL) { return -1; } }
  sc = (stream_ctx_t*)calloc(1, sizeof(*sc));
  if (sc == NULL) { return SLA_APIRESULT_NG; }
  sc->parent = e; sc->input = input; sc->n = n; sc->data = data; sc->data_size = data_size;
  sc->K = K; sc->L = L; sc->maxb = e->encode_param.max_num_block_samples;
  for (r = 0; r <= K; r++) { const uint64_t at = (uint64_t)r * piece; sc->nominal[r] = (at < n && r < K) ? (uint32_t)at : n; }
  sc->bounds[0] = 0; sc->bounds_known = 1;
  sc->off[0] = SLA_HEADER_SIZE; sc->off_known = 1;
  pthread_mutex_init(&sc->mu, NULL); pthread_cond_init(&sc->cv, NULL);
  sc->t0 = now_ms();
  for (t = 0; t < L; t++) {
    args[t].sc = sc; args[t].lane = t;
    if (pthread_create(&th[t], NULL, stream_lane_main, &args[t]) != 0) { stream_fail(sc, SLA_APIRESULT_NG); break; }
    started++;
  }
  for (t = 0; t < started; t++) { pthread_join(th[t], NULL); }
  (void)enter(e);                                         /* the lanes named their own knobs on their threads; this thread is ours */
  rc = sc->failed;
  if (e->trace) {
    for (r = 0; r < K; r++) {
      fprintf(stderr, "[sla_hip] piece %2u lane %u: upload %7.3f..%7.3f scanned %7.3f hop %7.3f analysed %7.3f sized %7.3f delivered %7.3f ms\n", r, (uint32_t)sc->lane_of[r],
              sc->stamp[r][0], sc->stamp[r][1], sc->stamp[r][2], sc->stamp[r][3], sc->stamp[r][4], sc->stamp[r][5], sc->stamp[r][6]);
    }
  }
  if (rc == 0) {
    memset(&hinfo, 0, sizeof(hinfo));
    hinfo.wave_format = e->wave_format; hinfo.wave_format.offset_lshift = (uint8_t)sc->lshift;
    hinfo.encode_param = e->encode_param; hinfo.num_samples = n; hinfo.num_blocks = sc->num_blocks;
    hinfo.max_block_size = sc->max_block; hinfo.max_bit_per_second = sc->max_bps;
    rc = slai_write_header(&hinfo, data, data_size);
    if (rc == 0) { *output_size = (uint32_t)sc->off[K]; e->streamed = 1; e->analysed = 0; e->lshift = sc->lshift; e->num_samples = n; }
  }
  pthread_mutex_destroy(&sc->mu); pthread_cond_destroy(&sc->cv);
  free(sc);
  return rc;
}

SLAApiResult SLAEncoder_EncodeWhole(struct SLAEncoder* e, const int32_t* const* input, uint32_t num_samples,
                                    uint8_t* data, uint32_t data_size, uint32_t* output_size)
{
  int rc;
  if (e == NULL || input == NULL || data == NULL || output_size == NULL) { return SLA_APIRESULT_INVALID_ARGUMENT; }
  if ((rc = check_ready(e)) != 0) { return (SLAApiResult)rc; }
  if (data_size < SLA_HEADER_SIZE) { return SLA_APIRESULT_INSUFFICIENT_BUFFER_SIZE; }
  if (enter(e) != 0) { return SLA_APIRESULT_NG; }
  {
    const double t0 = now_ms();
    double t1, t2;
    e->streamed = 0;
    rc = encode_whole_streamed(e, input, num_samples, data, data_size, output_size);
    if (rc >= 0) {
      if (e->trace) { fprintf(stderr, "[sla_hip] EncodeWhole: streamed, %.3f ms\n", now_ms() - t0); }
      return (SLAApiResult)rc;
    }
    if ((rc = upload_pcm(e, input, num_samples)) != 0) { return (rc > 0) ? (SLAApiResult)rc : SLA_APIRESULT_NG; }
    t1 = now_ms();
    rc = sla_hip_analyze_device(e, e->pcm_dev, e->stride, num_samples, NULL, NULL);
    t2 = now_ms();
    if (rc == 0) { rc = sla_hip_pack_device(e, data, data_size, output_size); }
    if (e->trace) { fprintf(stderr, "[sla_hip] EncodeWhole: upload %.3f ms, analysis %.3f ms, pack + download %.3f ms\n", t1 - t0, t2 - t1, now_ms() - t2); }
  }
  return (rc >= 0) ? (SLAApiResult)rc : SLA_APIRESULT_NG;
}

/* ---- many files in one pass (BASELINE C4: a batch of short clips) --------------------------------------------
 * A 10-second clip is 235 blocks: far too few to fill the chip, and every stage of the pipeline is a kernel whose
 * duration is set by ONE block's serial chain, so a clip costs the same few milliseconds as a file a hundred times
 * longer.  Blocks of different files are as independent as blocks of one file, so the batch lays the
 * files out back to back in one set of planes (starts on 1024-sample tile boundaries, gaps zero), restarts the
 * super-frame hop at every file, and runs the pipeline ONCE; offset_lshift is per file (OR of the file's prepass
 * tiles) -- files that share a value share a pass.  Every file comes out byte-identical to its own
 * SLAEncoder_EncodeWhole. */
typedef struct {
  const sla_hip_batch_item* items; const uint32_t* start;      /* plane position of every file of the pass */
  uint32_t first, count;                                        /* items [first, first + count) */
  uint32_t ch; size_t plane_lo;                                 /* this slot covers plane samples [plane_lo, plane_lo + n) */
  size_t n; int16_t* dst16; int32_t* dst32; uint32_t lowbits;
} stage_batch_t;

static void stage_batch_one(void* vctx, uint32_t g)
{
  stage_batch_t* c = (stage_batch_t*)vctx;
  const size_t lo = (size_t)g * XFER_GRAIN, hi = (lo + XFER_GRAIN < c->n) ? lo + XFER_GRAIN : c->n;
  const size_t p_lo = c->plane_lo + lo, p_hi = c->plane_lo + hi;
  uint32_t a = 0, b = c->count, i, low = 0;
  size_t at = p_lo;
  /* first file whose end lies behind p_lo (starts ascend) */
  while (a < b) { const uint32_t m = (a + b) / 2; if ((size_t)c->start[m] + c->items[c->first + m].num_samples <= p_lo) { a = m + 1; } else { b = m; } }
  for (i = a; at < p_hi; i++) {
    const size_t f_lo = (i < c->count) ? c->start[i] : p_hi, f_hi = (i < c->count) ? f_lo + c->items[c->first + i].num_samples : p_hi;
    const size_t z_hi = (f_lo < p_hi) ? ((f_lo > at) ? f_lo : at) : p_hi;      /* gap [at, z_hi): zeros */
    size_t k;
    if (c->dst16 != NULL) { memset(c->dst16 + (at - c->plane_lo), 0, (z_hi - at) * 2); } else { memset(c->dst32 + (at - c->plane_lo), 0, (z_hi - at) * 4); }
    at = z_hi;
    if (at >= p_hi || i >= c->count) { break; }
    {
      const size_t c_hi = (f_hi < p_hi) ? f_hi : p_hi;
      const int32_t* src = c->items[c->first + i].input[c->ch] + (at - f_lo);
      if (c_hi <= at) { continue; }                                            /* a file of zero samples */
      if (c->dst16 != NULL) {
        int16_t* d = c->dst16 + (at - c->plane_lo);
        for (k = 0; k < c_hi - at; k++) { const int32_t v = src[k]; low |= (uint32_t)v & 0xFFFFu; d[k] = (int16_t)(v >> 16); }
      } else {
        memcpy(c->dst32 + (at - c->plane_lo), src, sizeof(int32_t) * (c_hi - at));
      }
      at = c_hi;
    }
  }
  if (low) { __atomic_fetch_or(&c->lowbits, low, __ATOMIC_RELAXED); }
}

static int upload_batch_pass(struct SLAEncoder* e, const sla_hip_batch_item* items, const uint32_t* start, uint32_t first,
                             uint32_t count, size_t span, uint64_t stride, int mode16, uint32_t* lowbits)
{
  extern int sla_hip_launch_unpack16(const int16_t*, int32_t*, uint64_t, sla_hip_stream_t);
  const uint32_t C = e->wave_format.num_channels;
  const size_t slot_samples = mode16 ? (XFER_SLOT_BYTES / 2) : (XFER_SLOT_BYTES / 4);
  uint32_t ch, k = 0;
  size_t o;
  stage_batch_t ctx;
  ctx.items = items; ctx.start = start; ctx.first = first; ctx.count = count; ctx.lowbits = 0;
  for (ch = 0; ch < C; ch++) {
    for (o = 0; o < span; o += slot_samples, k++) {
      const uint32_t slot = k & 1;
      const size_t n = (span - o < slot_samples) ? (span - o) : slot_samples;
      int32_t* dst = (int32_t*)e->d_pcm.ptr + (size_t)ch * stride + o;
      if (k >= 2) { HIPCHK(hipEventSynchronize(e->ev_stage[slot])); }
      ctx.ch = ch; ctx.plane_lo = o; ctx.n = n;
      ctx.dst16 = mode16 ? (int16_t*)e->h_stage[slot].ptr : NULL;
      ctx.dst32 = mode16 ? NULL : (int32_t*)e->h_stage[slot].ptr;
      parallel_for(e->upload_pool != NULL ? e->upload_pool : e->pool, (uint32_t)((n + XFER_GRAIN - 1) / XFER_GRAIN), stage_batch_one, &ctx);
      if (mode16) {
        HIPCHK(hipMemcpyAsync(e->d_stage[slot].ptr, e->h_stage[slot].ptr, n * 2, hipMemcpyHostToDevice, e->stream));
        RCCHK(sla_hip_launch_unpack16((const int16_t*)e->d_stage[slot].ptr, dst, n, e->stream));
      } else {
        HIPCHK(hipMemcpyAsync(dst, e->h_stage[slot].ptr, n * 4, hipMemcpyHostToDevice, e->stream));
      }
      HIPCHK(hipEventRecord(e->ev_stage[slot], e->stream));
    }
  }
  HIPCHK(hipStreamSynchronize(e->stream));
  *lowbits = ctx.lowbits;
  return 0;
}

#define BATCH_MAX_SPAN (1u << 28)        /* samples per channel in one pass: C x 1 GiB of planes (x3 with the residuals) */

/* Prepass over planes that hold `count` files back to back: the silence mask of everything on the host, and per file
 * the OR of its samples (from the 1024-sample tiles) and its offset_lshift (src/SLAEncoder.c:425-455; 0xFFFFFFFF:
 * samples wider than the declared depth). */
static int batch_prepass(struct SLAEncoder* e, uint64_t span, const uint32_t* start, const uint32_t* len, uint32_t count,
                         uint32_t* lsh, uint32_t* orv)
{
  const uint32_t C = e->wave_format.num_channels, bps = e->wave_format.bit_per_sample;
  const uint32_t ms = (e->encode_param.ch_process_method == SLA_CHPROCESSMETHOD_STEREO_MS);
  const uint64_t nwords = (span + 63) / 64;
  const uint32_t ntiles = (uint32_t)((span + 4 * SLA_HIP_PREPASS_TILE - 1) / (4 * SLA_HIP_PREPASS_TILE) * 4);
  uint32_t i;
  RCCHK(dev_reserve(&e->d_or, 64));
  RCCHK(dev_reserve(&e->d_nz, (size_t)(nwords + 2) * 8));
  RCCHK(dev_reserve(&e->d_tile_or, sizeof(uint32_t) * ((size_t)ntiles + 4)));
  RCCHK(dev_reserve(&e->d_binfo, sizeof(uint32_t) * 5 * (size_t)count + 64));
  RCCHK(pin_reserve(&e->h_binfo, sizeof(uint32_t) * 5 * (size_t)count + 64));
  {
    /* the files' positions go up behind nothing, the prepass and the per-file summary (k_batch_scan) follow, and what
     * comes home is 12 bytes per file: its OR word, its all-zero mask words, "its last super-frame is silent".  Only a
     * batch with silence in it brings the whole mask home (N/8 bytes: 7.5 MB and 0.3 ms for 125 ten-second clips) */
    uint32_t* hb = (uint32_t*)e->h_binfo.ptr;
    const uint32_t* info = hb + 2 * (size_t)count;
    int silence = 0;
    memcpy(hb, start, sizeof(uint32_t) * count); memcpy(hb + count, len, sizeof(uint32_t) * count);
    HIPCHK(hipMemcpyAsync(e->d_binfo.ptr, hb, sizeof(uint32_t) * 2 * (size_t)count, hipMemcpyHostToDevice, e->stream));
    HIPCHK(hipEventRecord(e->ev[0], e->stream));
    RCCHK(sla_hip_launch_prepass_tiles(e->pcm_dev, e->stride, C, (uint32_t)span, bps, ms, (uint32_t*)e->d_or.ptr, (uint64_t*)e->d_nz.ptr,
                                       (uint32_t*)e->d_tile_or.ptr, e->stream));
    HIPCHK(hipEventRecord(e->ev[1], e->stream));
    RCCHK(sla_hip_launch_batch_scan((const uint64_t*)e->d_nz.ptr, (const uint32_t*)e->d_tile_or.ptr, (const uint32_t*)e->d_binfo.ptr,
                                    (const uint32_t*)e->d_binfo.ptr + count, count, e->encode_param.max_num_block_samples,
                                    (uint32_t*)e->d_binfo.ptr + 2 * (size_t)count, e->stream));
    HIPCHK(hipMemcpyAsync(hb + 2 * (size_t)count, (uint32_t*)e->d_binfo.ptr + 2 * (size_t)count, sizeof(uint32_t) * 3 * (size_t)count,
                          hipMemcpyDeviceToHost, e->stream));
    HIPCHK(hipStreamSynchronize(e->stream));
    for (i = 0; i < count; i++) { if (info[3 * i + 1] != 0 || info[3 * i + 2] != 0) { silence = 1; } }
    e->batch_silence = silence;
    if (silence) {
      RCCHK(pin_reserve(&e->h_nz, (size_t)(nwords + 2) * 8));
      HIPCHK(hipMemcpyAsync(e->h_nz.ptr, e->d_nz.ptr, (size_t)nwords * 8, hipMemcpyDeviceToHost, e->stream));
      HIPCHK(hipStreamSynchronize(e->stream));
      ((uint64_t*)e->h_nz.ptr)[nwords] = 0; ((uint64_t*)e->h_nz.ptr)[nwords + 1] = 0;
      e->nz_ones_words = 0;                               /* the single-file path may not assume anything about h_nz any more */
    }
    for (i = 0; i < count; i++) {
      const uint32_t mask = info[3 * i];
      orv[i] = mask; lsh[i] = 0;
      if (mask != 0) {
        const uint32_t ntz = (uint32_t)__builtin_ctz(mask);
        lsh[i] = (bps < 32 - ntz || bps - (32 - ntz) >= bps) ? 0xFFFFFFFFu : bps - (32 - ntz);
      }
    }
  }
  return 0;
}

/* items [first, first + count): one upload, one prepass, one pipeline pass per distinct offset_lshift */
static int encode_batch_pass(struct SLAEncoder* e, sla_hip_batch_item* items, uint32_t first, uint32_t count)
{
  const uint32_t C = e->wave_format.num_channels, bps = e->wave_format.bit_per_sample;
  uint32_t *start, *lsh, *seg_start, *seg_len, *seg_item, *orv;
  pack_seg_t* segs;
  uint32_t i, lowbits = 0;
  uint64_t span = 0, stride;
  int mode16 = (bps <= 16), rc = 0;

  start = (uint32_t*)malloc(sizeof(uint32_t) * (size_t)count * 6 + 64);
  segs = (pack_seg_t*)malloc(sizeof(pack_seg_t) * ((size_t)count + 1));
  if (start == NULL || segs == NULL) { free(start); free(segs); return SLA_APIRESULT_NG; }
  lsh = start + count; seg_start = lsh + count; seg_len = seg_start + count; seg_item = seg_len + count; orv = seg_item + count;
  for (i = 0; i < count; i++) {
    start[i] = (uint32_t)span;
    span += ((uint64_t)items[first + i].num_samples + SLA_HIP_PREPASS_TILE - 1) / SLA_HIP_PREPASS_TILE * SLA_HIP_PREPASS_TILE;
  }
  if (span == 0) { span = SLA_HIP_PREPASS_TILE; }
  stride = span;
#define BATCH_CHK(call) do { rc = (call); if (rc != 0) { goto done; } } while (0)
#define BATCH_HIP(call) do { if ((call) != hipSuccess) { rc = SLA_APIRESULT_NG; goto done; } } while (0)
  BATCH_CHK(dev_reserve(&e->d_pcm, sizeof(int32_t) * (size_t)C * (stride + 64)));
  for (i = 0; i < 2; i++) { BATCH_CHK(pin_reserve(&e->h_stage[i], XFER_SLOT_BYTES)); BATCH_CHK(dev_reserve(&e->d_stage[i], XFER_SLOT_BYTES)); }
  if (e->upload_gate != NULL) { pthread_mutex_lock(e->upload_gate); }
  e->batch_stamp[0] = now_ms();
  rc = upload_batch_pass(e, items, start, first, count, (size_t)span, stride, mode16, &lowbits);
  if (rc == 0 && mode16 && lowbits != 0) { rc = upload_batch_pass(e, items, start, first, count, (size_t)span, stride, 0, &lowbits); }
  e->batch_stamp[1] = now_ms();
  if (e->upload_gate != NULL) { pthread_mutex_unlock(e->upload_gate); }
  if (rc != 0) { goto done; }
  e->pcm_dev = (const int32_t*)e->d_pcm.ptr; e->stride = stride; e->num_samples = (uint32_t)span;

  /* prepass over everything: silence mask, offset_lshift per file */
  for (i = 0; i < count; i++) { seg_len[i] = items[first + i].num_samples; }
  BATCH_CHK(batch_prepass(e, span, start, seg_len, count, lsh, orv));
  for (i = 0; i < count; i++) {
    items[first + i].result = SLA_APIRESULT_OK; items[first + i].output_size = 0;
    if (lsh[i] == 0xFFFFFFFFu) { items[first + i].result = SLA_APIRESULT_INVALID_ARGUMENT; continue; }
    if (items[first + i].data == NULL || items[first + i].data_size < SLA_HEADER_SIZE) {
      items[first + i].result = (items[first + i].data == NULL) ? SLA_APIRESULT_INVALID_ARGUMENT : SLA_APIRESULT_INSUFFICIENT_BUFFER_SIZE;
      lsh[i] = 0xFFFFFFFFu;
    }
  }

  /* one pipeline pass per distinct offset_lshift (normally one) */
  for (;;) {
    uint32_t v = 0xFFFFFFFFu, ns = 0, s, gor = 0;
    for (i = 0; i < count; i++) { if (lsh[i] != 0xFFFFFFFFu) { v = lsh[i]; break; } }
    if (v == 0xFFFFFFFFu) { break; }
    for (i = 0; i < count; i++) {
      if (lsh[i] != v) { continue; }
      seg_start[ns] = start[i]; seg_len[ns] = items[first + i].num_samples; seg_item[ns] = first + i; gor |= orv[i];
      lsh[i] = 0xFFFFFFFFu; ns++;
    }
    e->nsegs = ns; e->seg_start = seg_start; e->seg_len = seg_len; e->batch_lshift = v; e->batch_or = gor;
    e->analysed = 0;
    memset(e->timing, 0, sizeof(e->timing));
    rc = run_pipeline(e, 0);
    e->nsegs = 0; e->seg_start = NULL; e->seg_len = NULL;
    if (rc != 0) { goto done; }
    e->wave_format.offset_lshift = (uint8_t)e->lshift;
    e->analysed = 1;
    e->batch_stamp[2] = now_ms();
    for (s = 0; s < ns; s++) {
      memset(&segs[s], 0, sizeof(segs[s]));
      segs[s].lo = seg_start[s]; segs[s].hi = seg_start[s] + seg_len[s];
      segs[s].data = items[seg_item[s]].data; segs[s].data_size = items[seg_item[s]].data_size;
    }
    /* a file of zero samples owns no tile: give it an empty range behind its predecessor so that the ranges stay sorted */
    rc = pack_device_core(e, segs, ns);
    if (rc != 0) { goto done; }
    for (s = 0; s < ns; s++) { items[seg_item[s]].result = segs[s].result; items[seg_item[s]].output_size = (segs[s].result == 0) ? segs[s].out_size : 0; }
  }
done:
  e->batch_stamp[3] = now_ms();
  e->nsegs = 0; e->seg_start = NULL; e->seg_len = NULL;
  e->analysed = 0;                                        /* the planes hold a batch, not a file: no trace / pack afterwards */
  free(start); free(segs);
  return rc;
#undef BATCH_CHK
#undef BATCH_HIP
}

/* The hot path over a batch that already lives in device memory (what bench.py --config C4 times). */
int sla_hip_analyze_batch_device(struct SLAEncoder* e, const int32_t* d_pcm, uint64_t plane_stride, uint32_t span,
                                 const uint32_t* file_start, const uint32_t* file_samples, uint32_t num_files,
                                 uint32_t* file_lshift, float* timing_ms)
{
  const double t_start = now_ms();
  uint32_t *lsh, *orv, *seg_start, *seg_len;
  uint32_t i, passes = 0;
  float acc[12];
  int rc = 0;
  if (e == NULL || d_pcm == NULL || (num_files != 0 && (file_start == NULL || file_samples == NULL))) { return SLA_APIRESULT_INVALID_ARGUMENT; }
  RCCHK(check_ready(e));
  if (plane_stride < span || span > BATCH_MAX_SPAN) { return SLA_APIRESULT_INVALID_ARGUMENT; }
  for (i = 0; i < num_files; i++) {
    if (file_start[i] % SLA_HIP_PREPASS_TILE != 0 || (uint64_t)file_start[i] + file_samples[i] > span
        || (i > 0 && (uint64_t)file_start[i - 1] + file_samples[i - 1] > file_start[i])) { return SLA_APIRESULT_INVALID_ARGUMENT; }
  }
  e->analysed = 0;
  if (num_files == 0 || span == 0) { return 0; }
  RCCHK(enter(e));
  lsh = (uint32_t*)malloc(sizeof(uint32_t) * (size_t)num_files * 4);
  if (lsh == NULL) { return SLA_APIRESULT_NG; }
  orv = lsh + num_files; seg_start = orv + num_files; seg_len = seg_start + num_files;
  e->pcm_dev = d_pcm; e->stride = plane_stride; e->num_samples = span;
  memset(acc, 0, sizeof(acc));
  rc = batch_prepass(e, span, file_start, file_samples, num_files, lsh, orv);
  for (i = 0; rc == 0 && i < num_files; i++) {
    if (file_lshift != NULL) { file_lshift[i] = lsh[i]; }
    if (lsh[i] == 0xFFFFFFFFu) { rc = SLA_APIRESULT_INVALID_ARGUMENT; }
  }
  while (rc == 0) {
    uint32_t v = 0xFFFFFFFFu, ns = 0, gor = 0, k;
    for (i = 0; i < num_files; i++) { if (lsh[i] != 0xFFFFFFFFu) { v = lsh[i]; break; } }
    if (v == 0xFFFFFFFFu) { break; }
    for (i = 0; i < num_files; i++) {
      if (lsh[i] != v) { continue; }
      seg_start[ns] = file_start[i]; seg_len[ns] = file_samples[i]; gor |= orv[i]; lsh[i] = 0xFFFFFFFFu; ns++;
    }
    e->nsegs = ns; e->seg_start = seg_start; e->seg_len = seg_len; e->batch_lshift = v; e->batch_or = gor;
    memset(e->timing, 0, sizeof(e->timing));
    rc = run_pipeline(e, 0);
    e->nsegs = 0; e->seg_start = NULL; e->seg_len = NULL;
    for (k = 0; k < 12; k++) { acc[k] = (k == 9 || k == 11) ? e->timing[k] : acc[k] + e->timing[k]; }
    passes++;
  }
  free(lsh);
  if (rc != 0) { return rc; }
  memcpy(e->timing, acc, sizeof(acc));
  e->timing[7] = (float)(now_ms() - t_start);
  e->wave_format.offset_lshift = (uint8_t)e->lshift;
  if (timing_ms != NULL) { memcpy(timing_ms, e->timing, sizeof(e->timing)); }
  e->analysed = (passes == 1);          /* one pass: the block table of the whole batch is there for sla_hip_get_trace */
  return 0;
}

/* A big batch on worker lanes (round 4).  Done in one piece, the legs of sla_hip_encode_batch follow one another -- staging and
 * upload of every file, analysis, Rice walk + bit-pack, download of every file: 25.7 ms for 125 ten-second stereo clips whose
 * analysis takes 3.1 -- with the bus idle under the kernels and the device idle under the bus.  Files are independent, so the
 * batch is dealt out in groups of consecutive files to lanes (handles of their own on the same device, one host thread each;
 * the lanes of the streamed SLAEncoder_EncodeWhole): a lane that has delivered its group takes the next one, uploads take turns
 * (one mutex), and everything behind a group's upload -- kernels, pack, download -- runs beside the other lanes' uploads. */
typedef struct {
  struct SLAEncoder* parent; sla_hip_batch_item* items;
  uint32_t ngroups, next; uint32_t lo[4 * SLAI_STREAM_LANES + 1];
  pthread_mutex_t mu, gate; int failed;
  double t0, stamp[4 * SLAI_STREAM_LANES][5]; uint8_t lane_of[4 * SLAI_STREAM_LANES];
} batch_ctx_t;
typedef struct { batch_ctx_t* bc; uint32_t lane; } batch_arg_t;

static void* batch_lane_main(void* varg)
{
  batch_arg_t* ba = (batch_arg_t*)varg;
  batch_ctx_t* bc = ba->bc;
  struct SLAEncoder* l = bc->parent->lane[ba->lane];
  for (;;) {
    uint32_t g;
    int rc;
    pthread_mutex_lock(&bc->mu);
    g = bc->next++;
    if (bc->failed != 0) { g = bc->ngroups; }
    pthread_mutex_unlock(&bc->mu);
    if (g >= bc->ngroups) { break; }
    bc->stamp[g][0] = now_ms() - bc->t0;
    l->upload_gate = &bc->gate; l->upload_pool = bc->parent->pool;      /* (the gate also makes the parent's pool one lane's at a time) */
    rc = sla_hip_encode_batch(l, bc->items + bc->lo[g], bc->lo[g + 1] - bc->lo[g]);
    l->upload_gate = NULL; l->upload_pool = NULL;
    { int q; for (q = 0; q < 4; q++) { bc->stamp[g][q + 1] = l->batch_stamp[q] - bc->t0; } bc->lane_of[g] = (uint8_t)ba->lane; }
    if (rc != 0) {
      pthread_mutex_lock(&bc->mu);
      if (bc->failed == 0) { bc->failed = rc; }
      pthread_mutex_unlock(&bc->mu);
      break;
    }
  }
  return NULL;
}

/* 0: done; < 0: not taken (small batch, switched off, a lane could not be made) -- the plain path; > 0: API result */
static int encode_batch_on_lanes(struct SLAEncoder* e, sla_hip_batch_item* items, uint32_t num_items)
{
  const uint32_t C = e->wave_format.num_channels;
  batch_ctx_t* bc;
  batch_arg_t args[SLAI_STREAM_LANES];
  pthread_t th[SLAI_STREAM_LANES];
  uint64_t total = 0, acc = 0, target;
  uint32_t L, G, g, i, t, started = 0;
  int rc;
  if (!e->stream_mode || e->is_lane || e->batch_lanes < 2 || num_items < 8) { return -1; }
  for (i = 0; i < num_items; i++) { total += items[i].num_samples; }
  if (total * C < (16ull << 20)) { return -1; }                  /* (below that the per-group latencies outweigh the overlap) */
  L = e->batch_lanes;
  G = 2 * L;                                                     /* two groups per lane: the second round starts out of step */
  if (G > num_items / 2) { G = num_items / 2; }
  if (G < 2) { return -1; }
  if (L > G) { L = G; }
  for (t = 0; t < L; t++) { if (stream_lane(e, t) == NULL) { return -1; } }
  bc = (batch_ctx_t*)calloc(1, sizeof(*bc));
  if (bc == NULL) { return SLA_APIRESULT_NG; }
  bc->parent = e; bc->items = items;
  /* groups of consecutive files, about equal in samples */
  target = (total + G - 1) / G;
  bc->lo[0] = 0; g = 0;
  for (i = 0; i < num_items; i++) {
    acc += items[i].num_samples;
    if (acc >= target * (g + 1) && g + 1 < G && i + 1 < num_items) { bc->lo[++g] = i + 1; }
  }
  bc->lo[++g] = num_items; bc->ngroups = g;
  pthread_mutex_init(&bc->mu, NULL); pthread_mutex_init(&bc->gate, NULL);
  bc->t0 = now_ms();
  for (t = 0; t < L; t++) {
    args[t].bc = bc; args[t].lane = t;
    if (pthread_create(&th[t], NULL, batch_lane_main, &args[t]) != 0) {
      pthread_mutex_lock(&bc->mu); if (bc->failed == 0) { bc->failed = SLA_APIRESULT_NG; } pthread_mutex_unlock(&bc->mu);
      break;
    }
    started++;
  }
  for (t = 0; t < started; t++) { pthread_join(th[t], NULL); }
  (void)enter(e);                                         /* the lanes named their own knobs on their threads; this thread is ours */
  rc = bc->failed;
  if (e->trace) {
    for (g = 0; g < bc->ngroups; g++) {
      fprintf(stderr, "[sla_hip] group %2u (files %u..%u) lane %u: taken %7.3f upload %7.3f..%7.3f analysed %7.3f delivered %7.3f ms\n", g, bc->lo[g], bc->lo[g + 1],
              (uint32_t)bc->lane_of[g], bc->stamp[g][0], bc->stamp[g][1], bc->stamp[g][2], bc->stamp[g][3], bc->stamp[g][4]);
    }
  }
  pthread_mutex_destroy(&bc->mu); pthread_mutex_destroy(&bc->gate);
  free(bc);
  e->analysed = 0;
  return (rc > 0) ? rc : (rc < 0 ? SLA_APIRESULT_NG : 0);
}

int sla_hip_encode_batch(struct SLAEncoder* e, sla_hip_batch_item* items, uint32_t num_items)
{
  uint32_t first = 0, i, ch;
  int rc;
  if (e == NULL || (items == NULL && num_items != 0)) { return SLA_APIRESULT_INVALID_ARGUMENT; }
  if ((rc = check_ready(e)) != 0) { return rc; }
  RCCHK(enter(e));
  for (i = 0; i < num_items; i++) {
    if (items[i].input == NULL) { return SLA_APIRESULT_INVALID_ARGUMENT; }
    for (ch = 0; ch < e->wave_format.num_channels; ch++) { if (items[i].input[ch] == NULL) { return SLA_APIRESULT_INVALID_ARGUMENT; } }
    if (items[i].num_samples > BATCH_MAX_SPAN - SLA_HIP_PREPASS_TILE) { return SLA_APIRESULT_EXCEED_HANDLE_CAPACITY; }
  }
  rc = encode_batch_on_lanes(e, items, num_items);
  if (rc >= 0) { return rc; }
  while (first < num_items) {
    uint64_t span = 0;
    uint32_t count = 0;
    while (first + count < num_items) {
      const uint64_t add = ((uint64_t)items[first + count].num_samples + SLA_HIP_PREPASS_TILE - 1) / SLA_HIP_PREPASS_TILE * SLA_HIP_PREPASS_TILE;
      if (count > 0 && span + add > BATCH_MAX_SPAN) { break; }
      span += add; count++;
    }
    rc = encode_batch_pass(e, items, first, count);
    if (rc != 0) { return (rc > 0) ? rc : SLA_APIRESULT_NG; }
    first += count;
  }
  return 0;
}

/* One block with the encoder's current offset_lshift; no partition search (src/SLAEncoder.c:458-801). */
SLAApiResult SLAEncoder_EncodeBlock(struct SLAEncoder* e, const int32_t* const* input, uint32_t num_samples,
                                    uint8_t* data, uint32_t data_size, uint32_t* output_size)
{
  const int32_t* planes[SLAI_MAX_CHANNELS];
  uint8_t* tmp;
  uint32_t C, ch, size = 0, cap;
  uint64_t nwords;
  int rc;
  if (e == NULL || input == NULL || data == NULL || output_size == NULL) { return SLA_APIRESULT_INVALID_ARGUMENT; }
  if (!(e->status_flag & STATUS_WAVE_FORMAT) || !(e->status_flag & STATUS_ENCODE_PARAM)) { return SLA_APIRESULT_PARAMETER_NOT_SET; }
  if (num_samples > e->cfg.max_num_block_samples || num_samples > MAX_ANALYSIS_WINDOW) { return SLA_APIRESULT_EXCEED_HANDLE_CAPACITY; }
  if (data_size <= SLA_BLOCK_HEADER_SIZE) { return SLA_APIRESULT_INSUFFICIENT_DATA_SIZE; }
  if ((rc = check_ready(e)) != 0) { return (SLAApiResult)rc; }
  if (num_samples == 0) { return SLA_APIRESULT_INVALID_ARGUMENT; }
  C = e->wave_format.num_channels;
  if (enter(e) != 0) { return SLA_APIRESULT_NG; }
  if ((rc = upload_pcm(e, input, num_samples)) != 0) { return (rc > 0) ? (SLAApiResult)rc : SLA_APIRESULT_NG; }
  e->analysed = 0;
  e->lshift = e->wave_format.offset_lshift;
  memset(e->timing, 0, sizeof(e->timing));
  /* silence test through the prepass mask */
  nwords = ((uint64_t)num_samples + 63) / 64;
  if (dev_reserve(&e->d_or, 64) || dev_reserve(&e->d_nz, (size_t)(nwords + 2) * 8) || pin_reserve(&e->h_nz, (size_t)(nwords + 2) * 8)) { return SLA_APIRESULT_NG; }
  rc = sla_hip_launch_prepass(e->pcm_dev, e->stride, C, num_samples, e->wave_format.bit_per_sample,
                              e->encode_param.ch_process_method == SLA_CHPROCESSMETHOD_STEREO_MS,
                              (uint32_t*)e->d_or.ptr, (uint64_t*)e->d_nz.ptr, e->stream);
  if (rc != 0) { return SLA_APIRESULT_NG; }
  if (hipMemcpyAsync(e->h_nz.ptr, e->d_nz.ptr, (size_t)nwords * 8, hipMemcpyDeviceToHost, e->stream) != hipSuccess
      || hipStreamSynchronize(e->stream) != hipSuccess) { return SLA_APIRESULT_NG; }
  ((uint64_t*)e->h_nz.ptr)[nwords] = 0;
  e->nz_ones_words = 0;
  e->num_blocks = 0;
  if (blocks_push(e, 0, num_samples, slai_range_is_zero((const uint64_t*)e->h_nz.ptr, 0, num_samples) ? SLAI_BLK_SILENT : SLAI_BLK_COMPRESS) != 0) {
    return SLA_APIRESULT_NG;
  }
  rc = run_pipeline(e, 1);
  if (rc != 0) { return (rc > 0) ? (SLAApiResult)rc : SLA_APIRESULT_NG; }
  e->analysed = 1;
  /* pack into a scratch image (header + block) and hand back the block bytes only */
  cap = SLA_HEADER_SIZE + 64 + 16 * C * num_samples + 4096;
  tmp = (uint8_t*)malloc(cap);
  if (tmp == NULL) { return SLA_APIRESULT_NG; }
  for (ch = 0; ch < C; ch++) { planes[ch] = input[ch]; }
  rc = pack_impl(e, planes, tmp, cap, &size);
  if (rc == 0) {
    size -= SLA_HEADER_SIZE;
    if (size > data_size) { rc = SLA_APIRESULT_INSUFFICIENT_DATA_SIZE; }
    else { memcpy(data, tmp + SLA_HEADER_SIZE, size); *output_size = size; }
  }
  free(tmp);
  return (rc >= 0) ? (SLAApiResult)rc : SLA_APIRESULT_NG;
}

/* ------------------------------------------------------------------- introspection */

const int32_t* sla_hip_final_residual(const struct SLAEncoder* e, uint64_t* plane_stride)
{
  if (e == NULL || !e->analysed) { return NULL; }
  if (plane_stride != NULL) { *plane_stride = e->stride; }
  return RES2(e);
}

const int32_t* sla_hip_lattice_residual(const struct SLAEncoder* e, uint64_t* plane_stride)
{
  if (e == NULL || !e->analysed) { return NULL; }
  if (plane_stride != NULL) { *plane_stride = e->stride; }
  return RES1(e);
}

int sla_hip_bind_residual_planes(struct SLAEncoder* e, int32_t* d_lattice, int32_t* d_final, uint64_t plane_stride)
{
  if (e == NULL || ((d_lattice == NULL) != (d_final == NULL))) { return SLA_APIRESULT_INVALID_ARGUMENT; }
  e->user_res1 = d_lattice; e->user_res2 = d_final; e->user_stride = plane_stride;
  e->analysed = 0;
  return 0;
}

int sla_hip_get_trace(struct SLAEncoder* e, sla_hip_trace* tr)
{
  uint32_t C, O1, b, ch, t;
  if (e == NULL || tr == NULL) { return SLA_APIRESULT_INVALID_ARGUMENT; }
  if (!e->analysed) { return SLA_APIRESULT_PARAMETER_NOT_SET; }
  RCCHK(enter(e));
  C = e->wave_format.num_channels; O1 = e->encode_param.parcor_order + 1;
  if (tr->max_blocks < e->num_blocks || tr->order_stride < O1 || tr->sample_stride < e->num_samples
      || (tr->ltm_stride < e->encode_param.longterm_order)) { return SLA_APIRESULT_INSUFFICIENT_BUFFER_SIZE; }
  tr->num_blocks = e->num_blocks; tr->offset_lshift = e->lshift;
  for (b = 0; b < e->num_blocks; b++) {
    tr->blk_start[b] = e->blk[b].start; tr->blk_nsmpl[b] = e->blk[b].nsmpl;
    tr->blk_type[b] = e->blk[b].type; tr->blk_bytes[b] = e->blk[b].bytes;
    for (ch = 0; ch < C; ch++) {
      const size_t slot = (size_t)b * C + ch;
      const blkch_t* bc = &e->bc[slot];
      if (e->blk[b].type != SLAI_BLK_COMPRESS) { tr->rshift[slot] = 0; tr->pitch[slot] = 0; tr->rice_init[slot] = 0; continue; }
      for (t = 0; t < O1; t++) {
        tr->parcor[slot * tr->order_stride + t] = e->parcor[slot * O1 + t];
        tr->code[slot * tr->order_stride + t] = e->code[slot * O1 + t];
        tr->kint[slot * tr->order_stride + t] = e->kint[slot * O1 + t];
      }
      tr->rshift[slot] = bc->rshift; tr->pitch[slot] = bc->pitch; tr->rice_init[slot] = bc->rice_init;
      if (tr->parcor_exact != NULL) { tr->parcor_exact[slot] = e->parcor_exact[slot]; }
      for (t = 0; t < e->encode_param.longterm_order; t++) { tr->ltm_coef[slot * tr->ltm_stride + t] = bc->ltm_q[t]; }
    }
  }
  for (ch = 0; ch < C && e->num_samples > 0; ch++) {
    if (tr->res_lattice != NULL && RES1(e) != NULL) {
      HIPCHK(hipMemcpy(tr->res_lattice + (size_t)ch * tr->sample_stride, RES1(e) + (size_t)ch * e->stride,
                       sizeof(int32_t) * e->num_samples, hipMemcpyDeviceToHost));
    }
    if (tr->res_final != NULL && RES2(e) != NULL) {
      HIPCHK(hipMemcpy(tr->res_final + (size_t)ch * tr->sample_stride, RES2(e) + (size_t)ch * e->stride,
                       sizeof(int32_t) * e->num_samples, hipMemcpyDeviceToHost));
    }
  }
  return 0;
}
