/*
 * shard_rccl.c -- one file over the GPUs of a node from a plain C host: libsla_hip.so + RCCL, no Python.
 *
 * BASELINE.json's north star keeps the host in C and re-assembles the residual stream with a single RCCL all-gather
 * over xGMI.  This is that program: one process per GPU, the sequence of include/sla_hip.h ("one file, several GPUs"),
 * every exchange an ncclAllGather:
 *
 *   sla_hip_shard_scan_counts -> ncclAllGather of 3 x int32 per rank (OR word in two halves, zero-word count)
 *   [only when some rank counted silence: sla_hip_shard_scan -> ncclAllGather of the 1-bit mask pieces]
 *   sla_hip_shard_bounds      (host arithmetic, the same table on every rank)
 *   sla_hip_shard_analyze[_no_silence] on the rank's own super-frames    <- the hot path
 *   ncclAllGather of the final residual planes                           <- the north star's collective
 *   sla_hip_pack_device -> ncclAllGather of the image sizes, then of the (padded) images
 *   rank 0: sla_hip_shard_header + the blocks back to back = the .sla file; it must equal SLAEncoder_EncodeWhole of
 *   the whole file on one GPU byte for byte, and SLADecoder_DecodeWhole must return the input.
 *
 * Launch: RANK / WORLD_SIZE / LOCAL_RANK in the environment (as torchrun or mpirun export them; default 0 / 1 / 0);
 * world > 1 needs SLA_RCCL_ID_FILE, a path all ranks can reach: rank 0 writes the ncclUniqueId there.
 * Build: make -C sla_amd/csrc examples      Run (1 GPU): sla_amd/shard_rccl [seconds]
 */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <math.h>
#include <unistd.h>

#include <hip/hip_runtime_api.h>
#include <rccl/rccl.h>

#include "SLA.h"
#include "SLAEncoder.h"
#include "SLADecoder.h"
#include "sla_hip.h"

#define CHECK_HIP(x)  do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(2); } } while (0)
#define CHECK_NCCL(x) do { ncclResult_t r_ = (x); if (r_ != ncclSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, ncclGetErrorString(r_)); exit(2); } } while (0)
#define CHECK_SLA(x)  do { int r_ = (int)(x); if (r_ != 0) { fprintf(stderr, "%s:%d libsla_hip result %d\n", __FILE__, __LINE__, r_); exit(2); } } while (0)

enum { CHANNELS = 2, BITS = 16, RATE = 48000, ORDER = 16, LTM = 1, LMS = 8, MAX_BLOCK = 4096, PIECE_ALIGN = 1024 };

/* BASELINE.md's generator: three sines + LCG noise, rounded to BITS, left-justified */
static void synth(int32_t* plane, uint32_t ch, uint32_t n)
{
  uint32_t s = 12345u + 7919u * ch, i;
  const double full = (double)(1 << (BITS - 1));
  for (i = 0; i < n; i++) {
    const double t = (double)i / RATE;
    double x, q;
    s = s * 1664525u + 1013904223u;
    x = 0.35 * sin(2 * M_PI * 220.0 * (ch + 1) * t) + 0.2 * sin(2 * M_PI * 1333.7 * t + ch) + 0.1 * sin(2 * M_PI * 5011.3 * t)
      + (((double)(s >> 8) / 16777216.0) - 0.5) * 0.04;
    q = floor(x * full + 0.5);
    if (q > full - 1) { q = full - 1; }
    if (q < -full) { q = -full; }
    plane[i] = (int32_t)((uint32_t)(int32_t)q << (32 - BITS));
  }
}

static uint32_t cut(uint32_t n, uint32_t world, uint32_t r)        /* scan pieces: sla_amd/dist.py scan_piece */
{
  if (r >= world) { return n; }
  return (uint32_t)((((uint64_t)n * r + world - 1) / world) / PIECE_ALIGN * PIECE_ALIGN);
}

int main(int argc, char** argv)
{
  const uint32_t rank = getenv("RANK") ? (uint32_t)atoi(getenv("RANK")) : 0;
  const uint32_t world = getenv("WORLD_SIZE") ? (uint32_t)atoi(getenv("WORLD_SIZE")) : 1;
  const int local = getenv("LOCAL_RANK") ? atoi(getenv("LOCAL_RANK")) : 0;
  const uint32_t n = (uint32_t)(RATE * (argc > 1 ? atof(argv[1]) : 20.0));
  int ndev = 0;
  ncclUniqueId id;
  ncclComm_t comm;
  hipStream_t st;
  uint32_t r, ch;

  CHECK_HIP(hipGetDeviceCount(&ndev));
  CHECK_HIP(hipSetDevice(local % ndev));
  CHECK_HIP(hipStreamCreate(&st));
  if (world > 1) {
    const char* path = getenv("SLA_RCCL_ID_FILE");
    FILE* f;
    if (path == NULL) { fprintf(stderr, "world > 1 needs SLA_RCCL_ID_FILE\n"); return 2; }
    /* The file carries the launch's own nonce (SLA_RCCL_LAUNCH_ID: the launcher hands every rank of ONE launch the same
     * value, e.g. its pid or a timestamp) in front of the id: a file left behind by an earlier launch -- or by rank 0 of
     * this one before it renamed the new id into place -- is not ours and is not read (a stale file gave
     * the ranks different ids and ncclCommInitRank hung).  Rank 0 removes what is there first; the others give up after
     * a minute instead of polling for ever. */
    const char* nonce_s = getenv("SLA_RCCL_LAUNCH_ID");
    const unsigned long long nonce = nonce_s ? strtoull(nonce_s, NULL, 10) : 0ull;
    if (nonce_s == NULL) { fprintf(stderr, "world > 1 needs SLA_RCCL_LAUNCH_ID (one value per launch, the same on every rank)\n"); return 2; }
    if (rank == 0) {
      char tmp[4096];
      (void)remove(path);
      CHECK_NCCL(ncclGetUniqueId(&id));
      snprintf(tmp, sizeof(tmp), "%s.tmp", path);
      f = fopen(tmp, "wb");
      if (!f || fwrite(&nonce, sizeof(nonce), 1, f) != 1 || fwrite(&id, sizeof(id), 1, f) != 1) { return 2; }
      fclose(f);
      if (rename(tmp, path) != 0) { return 2; }
    } else {
      int tries = 0, got = 0;
      while (!got && tries++ < 6000) {
        unsigned long long seen = 0;
        f = fopen(path, "rb");
        if (f != NULL) {
          got = (fread(&seen, sizeof(seen), 1, f) == 1 && seen == nonce && fread(&id, sizeof(id), 1, f) == 1);
          fclose(f);
        }
        if (!got) { usleep(10000); }
      }
      if (!got) { fprintf(stderr, "rank %u: no id of launch %llu in %s after 60 s\n", rank, nonce, path); return 2; }
    }
  } else {
    CHECK_NCCL(ncclGetUniqueId(&id));
  }
  CHECK_NCCL(ncclCommInitRank(&comm, (int)world, id, (int)rank));

  /* the whole file on the host (every rank synthesises it; a real host would read its piece), the rank's piece on the device */
  int32_t* host[CHANNELS];
  for (ch = 0; ch < CHANNELS; ch++) { host[ch] = (int32_t*)malloc(sizeof(int32_t) * n); synth(host[ch], ch, n); }
  const uint32_t lo = cut(n, world, rank), hi = cut(n, world, rank + 1);
  uint32_t top = hi + PIECE_ALIGN - 1 + MAX_BLOCK;               /* include/sla_hip.h, step 1 */
  if (top > n) { top = n; }
  uint64_t stride = 0;                                           /* the same plane stride on every rank: the planes are gathered as they are */
  for (r = 0; r < world; r++) {
    uint32_t t = cut(n, world, r + 1) + PIECE_ALIGN - 1 + MAX_BLOCK;
    if (t > n) { t = n; }
    if ((uint64_t)(t - cut(n, world, r)) > stride) { stride = t - cut(n, world, r); }
  }
  stride = (stride + 63) / 64 * 64;
  int32_t *d_pcm, *d_lat, *d_fin, *d_all;
  CHECK_HIP(hipMalloc((void**)&d_pcm, sizeof(int32_t) * CHANNELS * stride));
  CHECK_HIP(hipMalloc((void**)&d_lat, sizeof(int32_t) * CHANNELS * stride));
  CHECK_HIP(hipMalloc((void**)&d_fin, sizeof(int32_t) * CHANNELS * stride));
  CHECK_HIP(hipMalloc((void**)&d_all, sizeof(int32_t) * CHANNELS * stride * world));
  CHECK_HIP(hipMemset(d_pcm, 0, sizeof(int32_t) * CHANNELS * stride));
  for (ch = 0; ch < CHANNELS; ch++) {
    CHECK_HIP(hipMemcpy(d_pcm + ch * stride, host[ch] + lo, sizeof(int32_t) * (top - lo), hipMemcpyHostToDevice));
  }

  struct SLAEncoderConfig cfg = { CHANNELS, MAX_BLOCK, ORDER, LTM, LMS, 0 };
  struct SLAWaveFormat wf = { CHANNELS, BITS, RATE, 0 };
  struct SLAEncodeParameter ep = { ORDER, LTM, LMS, SLA_CHPROCESSMETHOD_STEREO_MS, SLA_WINDOWFUNCTIONTYPE_SIN, MAX_BLOCK };
  struct SLAEncoder* enc = SLAEncoder_Create(&cfg);
  if (enc == NULL) { fprintf(stderr, "SLAEncoder_Create failed (no HIP device?)\n"); return 2; }
  CHECK_SLA(SLAEncoder_SetWaveFormat(enc, &wf));
  CHECK_SLA(SLAEncoder_SetEncodeParameter(enc, &ep));
  CHECK_SLA(sla_hip_bind_residual_planes(enc, d_lat, d_fin, stride));

  /* steps 1 + 2: OR word and zero-word count of every piece, 12 bytes per rank */
  uint32_t or_word = 0, zero_words = 0, file_or = 0, zeros = 0;
  int32_t h3[3], *d3, *d3all, *h3all = (int32_t*)malloc(sizeof(int32_t) * 3 * world);
  CHECK_SLA(sla_hip_shard_scan_counts(enc, d_pcm, stride, hi - lo, &or_word, &zero_words));
  h3[0] = (int32_t)(or_word & 0x7FFFFFFFu); h3[1] = (int32_t)(or_word >> 31); h3[2] = (int32_t)(zero_words > 0x7FFFFFFFu ? 0x7FFFFFFFu : zero_words);
  CHECK_HIP(hipMalloc((void**)&d3, sizeof(h3)));
  CHECK_HIP(hipMalloc((void**)&d3all, sizeof(int32_t) * 3 * world));
  CHECK_HIP(hipMemcpy(d3, h3, sizeof(h3), hipMemcpyHostToDevice));
  CHECK_NCCL(ncclAllGather(d3, d3all, 3, ncclInt32, comm, st));
  CHECK_HIP(hipStreamSynchronize(st));
  CHECK_HIP(hipMemcpy(h3all, d3all, sizeof(int32_t) * 3 * world, hipMemcpyDeviceToHost));
  for (r = 0; r < world; r++) { file_or |= (uint32_t)h3all[3 * r] | ((uint32_t)h3all[3 * r + 1] << 31); zeros += (uint32_t)h3all[3 * r + 2]; }

  /* step 3: bounds -- with the mask only when some piece contains an all-zero 64-sample word */
  uint32_t* bounds = (uint32_t*)malloc(sizeof(uint32_t) * (world + 1));
  uint64_t* mask = NULL;
  if (zeros != 0) {
    uint64_t pad = 1, total = ((uint64_t)n + 63) / 64, at = 0, *mine, *d_mine, *d_allm, *h_allm;
    for (r = 0; r < world; r++) { const uint64_t w = ((uint64_t)(cut(n, world, r + 1) - cut(n, world, r)) + 63) / 64; if (w > pad) { pad = w; } }
    mine = (uint64_t*)calloc(pad, 8); h_allm = (uint64_t*)malloc(8 * pad * world); mask = (uint64_t*)malloc(8 * (total + 2));
    CHECK_SLA(sla_hip_shard_scan(enc, d_pcm, stride, hi - lo, &or_word, mine));
    CHECK_HIP(hipMalloc((void**)&d_mine, 8 * pad)); CHECK_HIP(hipMalloc((void**)&d_allm, 8 * pad * world));
    CHECK_HIP(hipMemcpy(d_mine, mine, 8 * pad, hipMemcpyHostToDevice));
    CHECK_NCCL(ncclAllGather(d_mine, d_allm, pad, ncclUint64, comm, st));
    CHECK_HIP(hipStreamSynchronize(st));
    CHECK_HIP(hipMemcpy(h_allm, d_allm, 8 * pad * world, hipMemcpyDeviceToHost));
    for (r = 0; r < world; r++) {
      const uint64_t w = ((uint64_t)(cut(n, world, r + 1) - cut(n, world, r)) + 63) / 64;
      memcpy(mask + at, h_allm + (uint64_t)r * pad, 8 * w); at += w;
    }
    CHECK_HIP(hipFree(d_mine)); CHECK_HIP(hipFree(d_allm)); free(mine); free(h_allm);
  }
  CHECK_SLA(sla_hip_shard_bounds(n, MAX_BLOCK, mask, world, bounds));

  /* step 4: the hot path on the rank's own super-frames, then the north star's collective: the residual planes */
  const uint32_t own_lo = bounds[rank], own_hi = bounds[rank + 1];
  float timing[12];
  if (own_hi > own_lo) {
    if (mask == NULL && file_or != 0) { CHECK_SLA(sla_hip_shard_analyze_no_silence(enc, d_pcm + (own_lo - lo), stride, own_hi - own_lo, file_or, timing)); }
    else { CHECK_SLA(sla_hip_shard_analyze(enc, d_pcm + (own_lo - lo), stride, own_hi - own_lo, file_or, timing)); }
  }
  CHECK_NCCL(ncclAllGather(d_fin, d_all, CHANNELS * stride, ncclInt32, comm, st));
  CHECK_HIP(hipStreamSynchronize(st));

  /* step 5: the compressed images, sizes first */
  const uint32_t cap = 8u * CHANNELS * (own_hi - own_lo) + 65536u;
  uint8_t* image = (uint8_t*)malloc(cap);
  uint32_t image_size = 0;
  if (own_hi > own_lo) { CHECK_SLA(sla_hip_pack_device(enc, image, cap, &image_size)); }
  uint32_t *d_sz, *d_szall, *sizes = (uint32_t*)malloc(sizeof(uint32_t) * world), maxsz = 1;
  CHECK_HIP(hipMalloc((void**)&d_sz, 4)); CHECK_HIP(hipMalloc((void**)&d_szall, 4 * world));
  CHECK_HIP(hipMemcpy(d_sz, &image_size, 4, hipMemcpyHostToDevice));
  CHECK_NCCL(ncclAllGather(d_sz, d_szall, 1, ncclUint32, comm, st));
  CHECK_HIP(hipStreamSynchronize(st));
  CHECK_HIP(hipMemcpy(sizes, d_szall, 4 * world, hipMemcpyDeviceToHost));
  for (r = 0; r < world; r++) { if (sizes[r] > maxsz) { maxsz = sizes[r]; } }
  uint8_t *d_img, *d_imgall, *all = (uint8_t*)malloc((size_t)maxsz * world);
  CHECK_HIP(hipMalloc((void**)&d_img, maxsz)); CHECK_HIP(hipMalloc((void**)&d_imgall, (size_t)maxsz * world));
  CHECK_HIP(hipMemset(d_img, 0, maxsz));
  CHECK_HIP(hipMemcpy(d_img, image, image_size, hipMemcpyHostToDevice));
  CHECK_NCCL(ncclAllGather(d_img, d_imgall, maxsz, ncclUint8, comm, st));
  CHECK_HIP(hipStreamSynchronize(st));
  CHECK_HIP(hipMemcpy(all, d_imgall, (size_t)maxsz * world, hipMemcpyDeviceToHost));

  int rc = 0;
  if (rank == 0) {
    /* one header over the shards' headers, the blocks back to back */
    const uint8_t** headers = (const uint8_t**)malloc(sizeof(uint8_t*) * world);
    size_t total = SLA_HEADER_SIZE, at = SLA_HEADER_SIZE;
    uint32_t shards = 0;
    for (r = 0; r < world; r++) { if (sizes[r] >= SLA_HEADER_SIZE) { headers[shards++] = all + (size_t)r * maxsz; total += sizes[r] - SLA_HEADER_SIZE; } }
    uint8_t* file = (uint8_t*)malloc(total);
    CHECK_SLA(sla_hip_shard_header(headers, shards, file, (uint32_t)total));
    for (r = 0; r < world; r++) {
      if (sizes[r] >= SLA_HEADER_SIZE) { memcpy(file + at, all + (size_t)r * maxsz + SLA_HEADER_SIZE, sizes[r] - SLA_HEADER_SIZE); at += sizes[r] - SLA_HEADER_SIZE; }
    }
    /* must equal the single-GPU encode of the whole file, and decode back to the input */
    struct SLAEncoder* whole = SLAEncoder_Create(&cfg);
    const int32_t* planes[CHANNELS];
    uint8_t* ref = (uint8_t*)malloc(8u * CHANNELS * (size_t)n + 65536u);
    uint32_t ref_size = 0, got = 0;
    for (ch = 0; ch < CHANNELS; ch++) { planes[ch] = host[ch]; }
    CHECK_SLA(SLAEncoder_SetWaveFormat(whole, &wf));
    CHECK_SLA(SLAEncoder_SetEncodeParameter(whole, &ep));
    CHECK_SLA(SLAEncoder_EncodeWhole(whole, planes, n, ref, 8u * CHANNELS * n + 65536u, &ref_size));
    if (ref_size != total || memcmp(ref, file, total) != 0) { fprintf(stderr, "sharded file differs from the single-GPU file\n"); rc = 1; }
    struct SLADecoderConfig dcfg = { CHANNELS, MAX_BLOCK, ORDER, LTM, LMS, 1, 0 };
    struct SLADecoder* dec = SLADecoder_Create(&dcfg);
    int32_t* back[CHANNELS];
    for (ch = 0; ch < CHANNELS; ch++) { back[ch] = (int32_t*)malloc(sizeof(int32_t) * n); }
    CHECK_SLA(SLADecoder_DecodeWhole(dec, file, (uint32_t)total, back, n, &got));
    for (ch = 0; ch < CHANNELS && rc == 0; ch++) { if (got != n || memcmp(back[ch], host[ch], sizeof(int32_t) * n) != 0) { fprintf(stderr, "round trip differs\n"); rc = 1; } }
    printf("{\"ranks\": %u, \"samples_per_channel\": %u, \"channels\": %d, \"sla_bytes\": %zu, \"silence_mask_exchanged\": %s, "
           "\"identical_to_single_gpu\": %s, \"round_trip\": %s, \"collectives\": \"ncclAllGather x %d\"}\n",
           world, n, (int)CHANNELS, total, mask ? "true" : "false", rc == 0 ? "true" : "false", rc == 0 ? "true" : "false", mask ? 5 : 4);
    SLADecoder_Destroy(dec); SLAEncoder_Destroy(whole);
  }
  SLAEncoder_Destroy(enc);
  CHECK_NCCL(ncclCommDestroy(comm));
  return rc;
}
