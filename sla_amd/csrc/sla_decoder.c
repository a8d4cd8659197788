/*
 * sla_decoder.c -- host side of the whole-file decoder (include/SLADecoder.h).
 *
 * The host reads 10 bytes per block (sync code, size field, sample count) to lay the block chain out as a
 * table; every other byte of the stream is consumed by the kernels of sla_decode.hip, all blocks at once.
 * Results are then examined in file order so that the first failing block decides the return code exactly
 * as the reference's block-by-block loop does (reference src/SLADecoder.c:660-732).
 * There is no CPU decode path in this file: without a HIP device SLADecoder_Create returns NULL.
 */
#include "sla_internal.h"
#include "SLADecoder.h"

#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#define DEC_MAX_BLOCK_SAMPLES 16384u
#define DEC_MIN_BLOCK_HEADER  11u        /* SLA_MINIMUM_BLOCK_HEADER_SIZE, reference src/include/private/SLAInternal.h:35 */
#define STATUS_WAVE_FORMAT    1u
#define STATUS_ENCODE_PARAM   2u

typedef struct { void* ptr; size_t cap; } dbuf_t;

struct SLADecoder {
  struct SLADecoderConfig   cfg;
  struct SLAWaveFormat      wave_format;
  struct SLAEncodeParameter encode_param;
  uint32_t                  status_flag;
  hipStream_t               stream;
  hipEvent_t                ev[2];
  dbuf_t                    d_image, d_planes, d_blocks, d_info, d_chan, d_kint;
  sla_hip_dec_block*        h_blocks;
  sla_hip_dec_info*         h_info;
  uint32_t                  h_cap;
  float                     timing[6];
};

static double now_ms(void)
{
  struct timespec ts;
  clock_gettime(CLOCK_MONOTONIC, &ts);
  return (double)ts.tv_sec * 1e3 + (double)ts.tv_nsec * 1e-6;
}

static int dbuf_reserve(dbuf_t* b, size_t bytes)
{
  if (bytes <= b->cap) { return 0; }
  if (b->ptr != NULL) { (void)hipFree(b->ptr); b->ptr = NULL; b->cap = 0; }
  bytes += bytes / 4 + 256;
  if (hipMalloc(&b->ptr, bytes) != hipSuccess) { b->ptr = NULL; return -1; }
  b->cap = bytes;
  return 0;
}
static void dbuf_free(dbuf_t* b) { if (b->ptr != NULL) { (void)hipFree(b->ptr); } b->ptr = NULL; b->cap = 0; }

static uint32_t rd_be16(const uint8_t* p) { return ((uint32_t)p[0] << 8) | p[1]; }
static uint32_t rd_be32(const uint8_t* p) { return ((uint32_t)p[0] << 24) | ((uint32_t)p[1] << 16) | ((uint32_t)p[2] << 8) | p[3]; }

/* reference src/SLADecoder.c:171-251 */
SLAApiResult SLADecoder_DecodeHeader(const uint8_t* data, uint32_t data_size, struct SLAHeaderInfo* header_info)
{
  struct SLAHeaderInfo h;
  SLAApiResult ret = SLA_APIRESULT_OK;
  if (data == NULL || header_info == NULL) { return SLA_APIRESULT_INVALID_ARGUMENT; }
  if (data_size < SLA_HEADER_SIZE) { return SLA_APIRESULT_INSUFFICIENT_DATA_SIZE; }
  if (data[0] != 'S' || data[1] != 'L' || data[2] != '*' || data[3] != 1) { return SLA_APIRESULT_INVALID_HEADER_FORMAT; }
  if (rd_be16(data + 8) != slai_crc16(data + SLAI_HDR_CRC_START, SLA_HEADER_SIZE - SLAI_HDR_CRC_START)) {
    ret = SLA_APIRESULT_DETECT_DATA_CORRUPTION;          /* reported, but the fields are still delivered */
  }
  if (rd_be32(data + 10) != SLA_FORMAT_VERSION) { return SLA_APIRESULT_INVALID_HEADER_FORMAT; }
  memset(&h, 0, sizeof(h));
  h.wave_format.num_channels          = data[14];
  h.num_samples                       = rd_be32(data + 15);
  h.wave_format.sampling_rate         = rd_be32(data + 19);
  h.wave_format.bit_per_sample        = data[23];
  h.wave_format.offset_lshift         = data[24];
  h.encode_param.parcor_order         = data[25];
  h.encode_param.longterm_order       = data[26];
  h.encode_param.lms_order_per_filter = data[27];
  h.encode_param.ch_process_method    = (SLAChannelProcessMethod)data[28];
  h.num_blocks                        = rd_be32(data + 29);
  h.encode_param.max_num_block_samples = rd_be16(data + 33);
  h.max_block_size                    = rd_be32(data + 35);
  h.max_bit_per_second                = rd_be32(data + 39);
  *header_info = h;
  return ret;
}

struct SLADecoder* SLADecoder_Create(const struct SLADecoderConfig* config)
{
  struct SLADecoder* d;
  int ndev = 0;
  if (config == NULL) { return NULL; }
  if (config->max_num_channels == 0 || config->max_num_channels > SLAI_MAX_CHANNELS
      || config->max_num_block_samples > DEC_MAX_BLOCK_SAMPLES || config->max_parcor_order > SLAI_MAX_ORDER
      || config->max_longterm_order > SLAI_MAX_TAPS) { return NULL; }
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) {
    fprintf(stderr, "libsla_hip: no HIP device available -- the decode path has no CPU fallback\n");
    return NULL;
  }
  d = (struct SLADecoder*)calloc(1, sizeof(*d));
  if (d == NULL) { return NULL; }
  d->cfg = *config;
  if (hipStreamCreateWithFlags(&d->stream, hipStreamNonBlocking) != hipSuccess) { free(d); return NULL; }
  if (hipEventCreate(&d->ev[0]) != hipSuccess || hipEventCreate(&d->ev[1]) != hipSuccess) { (void)hipStreamDestroy(d->stream); free(d); return NULL; }
  return d;
}

void SLADecoder_Destroy(struct SLADecoder* d)
{
  if (d == NULL) { return; }
  (void)hipStreamSynchronize(d->stream);
  dbuf_free(&d->d_image); dbuf_free(&d->d_planes); dbuf_free(&d->d_blocks);
  dbuf_free(&d->d_info); dbuf_free(&d->d_chan); dbuf_free(&d->d_kint);
  if (d->h_blocks != NULL) { (void)hipHostFree(d->h_blocks); }
  if (d->h_info != NULL) { (void)hipHostFree(d->h_info); }
  (void)hipEventDestroy(d->ev[0]); (void)hipEventDestroy(d->ev[1]);
  (void)hipStreamDestroy(d->stream);
  free(d);
}

SLAApiResult SLADecoder_SetWaveFormat(struct SLADecoder* d, const struct SLAWaveFormat* wave_format)
{
  if (d == NULL || wave_format == NULL) { return SLA_APIRESULT_INVALID_ARGUMENT; }
  if (wave_format->num_channels > d->cfg.max_num_channels || wave_format->bit_per_sample > 32) { return SLA_APIRESULT_EXCEED_HANDLE_CAPACITY; }
  d->wave_format = *wave_format;
  d->status_flag |= STATUS_WAVE_FORMAT;
  return SLA_APIRESULT_OK;
}

SLAApiResult SLADecoder_SetEncodeParameter(struct SLADecoder* d, const struct SLAEncodeParameter* ep)
{
  if (d == NULL || ep == NULL) { return SLA_APIRESULT_INVALID_ARGUMENT; }
  if (ep->parcor_order > d->cfg.max_parcor_order || ep->longterm_order > d->cfg.max_longterm_order
      || ep->lms_order_per_filter > d->cfg.max_lms_order_per_filter
      || ep->max_num_block_samples > d->cfg.max_num_block_samples
      || ep->max_num_block_samples < SLAI_MIN_BLOCK) { return SLA_APIRESULT_EXCEED_HANDLE_CAPACITY; }
  d->encode_param = *ep;
  d->status_flag |= STATUS_ENCODE_PARAM;
  return SLA_APIRESULT_OK;
}

static int host_tables_reserve(struct SLADecoder* d, uint32_t blocks)
{
  sla_hip_dec_block* nb = NULL;
  sla_hip_dec_info* ni = NULL;
  uint32_t cap;
  if (blocks <= d->h_cap) { return 0; }
  cap = blocks + blocks / 2 + 64;
  if (hipHostMalloc((void**)&nb, sizeof(*nb) * cap, hipHostMallocDefault) != hipSuccess) { return -1; }
  if (hipHostMalloc((void**)&ni, sizeof(*ni) * cap, hipHostMallocDefault) != hipSuccess) { (void)hipHostFree(nb); return -1; }
  if (d->h_blocks != NULL) { memcpy(nb, d->h_blocks, sizeof(*nb) * d->h_cap); (void)hipHostFree(d->h_blocks); }
  if (d->h_info != NULL) { (void)hipHostFree(d->h_info); }
  d->h_blocks = nb; d->h_info = ni; d->h_cap = cap;
  return 0;
}

#define HIPCHK(call) do { if ((call) != hipSuccess) { return SLA_APIRESULT_NG; } } while (0)
#define RCCHK(call)  do { const int rc_ = (call); if (rc_ != 0) { return (rc_ > 0) ? (SLAApiResult)rc_ : SLA_APIRESULT_NG; } } while (0)

/* The decode proper.  `host_data` is always the stream in host memory (the walk reads it); the image in device
 * memory is either uploaded from it or supplied by the caller, the planes likewise are the handle's or the caller's. */
static SLAApiResult decode_run(struct SLADecoder* d, const uint8_t* data, uint32_t data_size, const uint32_t* d_image_user,
                               int32_t* d_planes_user, uint64_t stride_user, int32_t** buffer,
                               uint32_t buffer_num_samples, uint32_t* output_num_samples, uint32_t* one_block_size)
{
  /* one_block_size == NULL: a whole file (header, then blocks until header.num_samples samples are out);
   * otherwise `data` starts at a block's sync code and exactly that block is decoded with the handle's current
   * format / parameters (the streaming decoder's unit; reference SLADecoder_DecodeBlock, src/SLADecoder.c:583-657) */
  struct SLAHeaderInfo header;
  SLAApiResult ret, result = SLA_APIRESULT_OK;
  uint32_t C, total, order, ntaps, lms, ms, bps, lshift, cap_n;
  uint32_t off = SLA_HEADER_SIZE, pos = 0, done_samples = 0, batches = 0, ch;
  uint64_t stride;
  const uint32_t* d_image;
  int32_t* d_planes;
  int lms_ok;
  float kernel_ms = 0.0f;
  double t0 = now_ms(), t_up = 0.0, t_walk = 0.0, t1;

  memset(d->timing, 0, sizeof(d->timing));
  if (one_block_size == NULL) {
    if ((ret = SLADecoder_DecodeHeader(data, data_size, &header)) != SLA_APIRESULT_OK) { return ret; }
    if ((ret = SLADecoder_SetWaveFormat(d, &header.wave_format)) != SLA_APIRESULT_OK) { return ret; }
    if ((ret = SLADecoder_SetEncodeParameter(d, &header.encode_param)) != SLA_APIRESULT_OK) { return ret; }
  } else {
    if (!(d->status_flag & STATUS_WAVE_FORMAT) || !(d->status_flag & STATUS_ENCODE_PARAM)) { return SLA_APIRESULT_PARAMETER_NOT_SET; }
    memset(&header, 0, sizeof(header));
    header.wave_format = d->wave_format; header.encode_param = d->encode_param;
    header.num_samples = 1;               /* the walk stops behind the first block */
    off = 0; *one_block_size = 0;
  }
  C = header.wave_format.num_channels; total = header.num_samples;
  bps = header.wave_format.bit_per_sample; lshift = header.wave_format.offset_lshift;
  order = header.encode_param.parcor_order; ntaps = header.encode_param.longterm_order;
  lms = header.encode_param.lms_order_per_filter;
  ms = (header.encode_param.ch_process_method == SLA_CHPROCESSMETHOD_STEREO_MS) ? 1u : 0u;
  cap_n = d->cfg.max_num_block_samples;
  *output_num_samples = 0;
  if (total == 0) { return SLA_APIRESULT_OK; }
  if (ms && C != 2) { return SLA_APIRESULT_INVAILD_CHPROCESSMETHOD; }            /* src/SLADecoder.c:605-613 */
  if (C == 0 || bps == 0 || lshift >= bps) { return SLA_APIRESULT_INVALID_HEADER_FORMAT; }   /* the reference asserts (src/SLADecoder.c:541-542) */
  lms_ok = (lms == 4 || lms == 8 || lms == 16 || lms == 32);

  /* the stream image on the device */
  if (d_image_user != NULL) { d_image = d_image_user; }
  else {
    const size_t padded = ((size_t)data_size + 3) & ~(size_t)3;
    if (dbuf_reserve(&d->d_image, padded + 16) != 0) { return SLA_APIRESULT_NG; }
    HIPCHK(hipMemsetAsync((uint8_t*)d->d_image.ptr + (padded - 4), 0, 4, d->stream));
    HIPCHK(hipMemcpyAsync(d->d_image.ptr, data, data_size, hipMemcpyHostToDevice, d->stream));
    d_image = (const uint32_t*)d->d_image.ptr;
  }
  if (d_planes_user != NULL) { d_planes = d_planes_user; stride = stride_user; }
  else {
    const uint64_t want = (uint64_t)total + 65535u;
    stride = (buffer_num_samples < want) ? buffer_num_samples : want;
    if (stride == 0) { stride = 1; }
    if (dbuf_reserve(&d->d_planes, (size_t)stride * C * sizeof(int32_t)) != 0) { return SLA_APIRESULT_NG; }
    d_planes = (int32_t*)d->d_planes.ptr;
  }
  t_up = now_ms() - t0;

  for (;;) {
    uint32_t nb = 0, i, resync = 0, batch_off = off, batch_pos = pos;
    SLAApiResult walk_err = SLA_APIRESULT_OK;
    double tw = now_ms();
    /* ---- walk the chain from (off, pos)                                  src/SLADecoder.c:696-722 */
    while (batch_pos < total) {
      const uint8_t* p;
      uint32_t left, bsize, n, flags = 0;
      if (batch_off > data_size) { walk_err = SLA_APIRESULT_INSUFFICIENT_DATA_SIZE; break; }
      left = data_size - batch_off;
      if (left < DEC_MIN_BLOCK_HEADER) { walk_err = SLA_APIRESULT_INSUFFICIENT_DATA_SIZE; break; }
      p = data + batch_off;
      if (rd_be16(p) != SLAI_SYNC_CODE) { walk_err = SLA_APIRESULT_FAILED_TO_FIND_SYNC_CODE; break; }
      bsize = rd_be32(p + 2) + 6u;
      n = rd_be16(p + 8);
      if (bsize > left || bsize < SLAI_BLK_CRC_START) { walk_err = SLA_APIRESULT_INSUFFICIENT_DATA_SIZE; break; }   /* no CRC check on a clipped block (:343) */
      if (n > buffer_num_samples - batch_pos || n > cap_n) {
        /* the CRC of this block is still checked first (:343-352 precede :633-636): keep it, header only */
        walk_err = SLA_APIRESULT_INSUFFICIENT_BUFFER_SIZE;
        if (d->cfg.enable_crc_check != 1) { break; }
        flags = SLA_HIP_DEC_HEADER_ONLY;
      }
      if (host_tables_reserve(d, nb + 1) != 0) { return SLA_APIRESULT_NG; }
      d->h_blocks[nb].byte_off = batch_off; d->h_blocks[nb].byte_len = bsize; d->h_blocks[nb].smp_off = batch_pos;
      d->h_blocks[nb].num_samples = n; d->h_blocks[nb].flags = flags;
      nb++;
      if (flags != 0) { break; }
      batch_off += bsize; batch_pos += n;
    }
    t_walk += now_ms() - tw;
    batches++;

    /* ---- all blocks of the batch through the kernels */
    if (nb > 0) {
      float ms_batch = 0.0f;
      if (dbuf_reserve(&d->d_blocks, sizeof(sla_hip_dec_block) * nb) != 0 || dbuf_reserve(&d->d_info, sizeof(sla_hip_dec_info) * nb) != 0
          || dbuf_reserve(&d->d_chan, sizeof(sla_hip_dec_chan) * (size_t)nb * C) != 0
          || dbuf_reserve(&d->d_kint, sizeof(int32_t) * (size_t)nb * C * (order + 1)) != 0) { return SLA_APIRESULT_NG; }
      HIPCHK(hipMemcpyAsync(d->d_blocks.ptr, d->h_blocks, sizeof(sla_hip_dec_block) * nb, hipMemcpyHostToDevice, d->stream));
      HIPCHK(hipMemsetAsync(d->d_info.ptr, 0, sizeof(sla_hip_dec_info) * nb, d->stream));
      HIPCHK(hipEventRecord(d->ev[0], d->stream));
      RCCHK(sla_hip_launch_dec_bits(d_image, data_size, (const sla_hip_dec_block*)d->d_blocks.ptr, nb, C, bps, lshift, ms, order, ntaps,
                                    d->cfg.enable_crc_check == 1, d_planes, stride, (sla_hip_dec_info*)d->d_info.ptr,
                                    (sla_hip_dec_chan*)d->d_chan.ptr, (int32_t*)d->d_kint.ptr, d->stream));
      HIPCHK(hipMemcpyAsync(d->h_info, d->d_info.ptr, sizeof(sla_hip_dec_info) * nb, hipMemcpyDeviceToHost, d->stream));
      if (lms_ok) {
        RCCHK(sla_hip_launch_dec_lms(d_planes, stride, (const sla_hip_dec_block*)d->d_blocks.ptr, (const sla_hip_dec_info*)d->d_info.ptr,
                                     nb, C, lms, d->stream));
        RCCHK(sla_hip_launch_dec_ltm(d_planes, stride, (const sla_hip_dec_block*)d->d_blocks.ptr, (const sla_hip_dec_info*)d->d_info.ptr,
                                     (const sla_hip_dec_chan*)d->d_chan.ptr, nb, C, ntaps, cap_n, d->stream));
        RCCHK(sla_hip_launch_dec_lattice(d_planes, stride, (const sla_hip_dec_block*)d->d_blocks.ptr, (const sla_hip_dec_info*)d->d_info.ptr,
                                         nb, C, (const int32_t*)d->d_kint.ptr, order, 1, d->stream));
      }
      HIPCHK(hipEventRecord(d->ev[1], d->stream));
      HIPCHK(hipStreamSynchronize(d->stream));
      if (hipEventElapsedTime(&ms_batch, d->ev[0], d->ev[1]) == hipSuccess) { kernel_ms += ms_batch; }
    }

    /* ---- examine the blocks in file order: the first failure decides */
    result = SLA_APIRESULT_OK;
    for (i = 0; i < nb; i++) {
      const sla_hip_dec_block* b = &d->h_blocks[i];
      const sla_hip_dec_info* in = &d->h_info[i];
      if (d->cfg.enable_crc_check == 1 && in->crc != rd_be16(data + b->byte_off + 6)) { result = SLA_APIRESULT_DETECT_DATA_CORRUPTION; break; }
      if (b->flags & SLA_HIP_DEC_HEADER_ONLY) { result = walk_err; break; }
      if (in->type > 2) { result = SLA_APIRESULT_INVALID_HEADER_FORMAT; break; }
      if (in->type == 0 && !lms_ok) { result = SLA_APIRESULT_FAILED_TO_SYNTHESIZE; break; }
      done_samples = b->smp_off + b->num_samples;
      if (one_block_size != NULL) { *one_block_size = in->used_bytes; break; }     /* src/SLADecoder.c:651 */
      if (in->used_bytes != b->byte_len) {
        /* the body did not end where the size field says: the reference continues from where its reader stopped (:715) */
        off = (uint32_t)b->byte_off + in->used_bytes; pos = done_samples; resync = 1;
        break;
      }
    }
    if (resync) { continue; }
    if (i == nb && result == SLA_APIRESULT_OK) { result = walk_err; }
    break;
  }

  /* ---- mid/side, left-justification, copy-out of everything before the failing block */
  if (done_samples > 0) {
    float ms_fin = 0.0f;
    HIPCHK(hipEventRecord(d->ev[0], d->stream));
    RCCHK(sla_hip_launch_dec_finish(d_planes, stride, C, done_samples, ms, 32u - bps + lshift, d->stream));
    HIPCHK(hipEventRecord(d->ev[1], d->stream));
    t1 = now_ms();
    if (buffer != NULL) {
      for (ch = 0; ch < C; ch++) {
        HIPCHK(hipMemcpyAsync(buffer[ch], d_planes + (uint64_t)ch * stride, sizeof(int32_t) * (size_t)done_samples, hipMemcpyDeviceToHost, d->stream));
      }
    }
    HIPCHK(hipStreamSynchronize(d->stream));
    if (hipEventElapsedTime(&ms_fin, d->ev[0], d->ev[1]) == hipSuccess) { kernel_ms += ms_fin; }
    d->timing[3] = (float)(now_ms() - t1);
  }
  *output_num_samples = done_samples;
  d->timing[0] = (float)t_up; d->timing[1] = (float)t_walk; d->timing[2] = kernel_ms;
  d->timing[4] = (float)(now_ms() - t0); d->timing[5] = (float)batches;
  return result;
}

SLAApiResult SLADecoder_DecodeWhole(struct SLADecoder* decoder, const uint8_t* data, uint32_t data_size,
                                    int32_t** buffer, uint32_t buffer_num_samples, uint32_t* output_num_samples)
{
  if (decoder == NULL || buffer == NULL || data == NULL || output_num_samples == NULL) { return SLA_APIRESULT_INVALID_ARGUMENT; }
  return decode_run(decoder, data, data_size, NULL, NULL, 0, buffer, buffer_num_samples, output_num_samples, NULL);
}

SLAApiResult sla_hip_decode_device(struct SLADecoder* decoder, const uint8_t* host_data, const uint32_t* d_image,
                                   uint32_t data_size, int32_t* d_planes, uint64_t plane_stride,
                                   uint32_t* output_num_samples)
{
  if (decoder == NULL || host_data == NULL || d_image == NULL || d_planes == NULL || output_num_samples == NULL) { return SLA_APIRESULT_INVALID_ARGUMENT; }
  return decode_run(decoder, host_data, data_size, d_image, d_planes, plane_stride, NULL,
                    (plane_stride > 0xFFFFFFFFull) ? 0xFFFFFFFFu : (uint32_t)plane_stride, output_num_samples, NULL);
}

int sla_hip_decoder_last_timing(const struct SLADecoder* decoder, float* timing_ms)
{
  if (decoder == NULL || timing_ms == NULL) { return SLA_APIRESULT_INVALID_ARGUMENT; }
  memcpy(timing_ms, decoder->timing, sizeof(decoder->timing));
  return 0;
}

/* One block that starts at data[0]: its samples, its sample count and the bytes it occupied. */
static SLAApiResult decode_one_block(struct SLADecoder* d, const uint8_t* data, uint32_t data_size, int32_t** buffer,
                                     uint32_t buffer_num_samples, uint32_t* block_size, uint32_t* num_samples)
{
  return decode_run(d, data, data_size, NULL, NULL, 0, buffer, buffer_num_samples, num_samples, block_size);
}

/* ------------------------------------------------------------------------------------------------------------
 * Streaming decoder (reference src/SLADecoder.c:735-1123): the caller appends fragments of the stream and draws
 * ceil(1.05 * sampling_rate / decode_interval_hz) samples per call.  Same entry points, same packet bookkeeping
 * (up to 8 fragments referenced in place, copied into a block buffer of twice the largest possible block as room
 * allows, handed back through CollectDataFragment).  The unit of device work is a whole block: a block is decoded
 * (one SLADecoder block decode on the device) as soon as all its bytes are in the buffer, and calls are served
 * from its samples.  Where the reference would start on a block whose tail has not arrived yet, this decoder
 * returns the samples it has (possibly none) with SLA_APIRESULT_OK and continues once the rest has been appended.
 * ------------------------------------------------------------------------------------------------------------ */
#define STREAM_MAX_PACKETS 8              /* SLA_STREAMING_DECODE_MAX_NUM_PACKETS */
#define STREAM_MARGIN      1.05f          /* SLA_STREAMING_DECODE_NUM_SAMPLES_MARGIN */

typedef struct { const uint8_t* data; uint32_t size, used; } packet_t;

struct SLAStreamingDecoder {
  struct SLADecoder* core;
  float     decode_interval_hz;
  uint32_t  max_bit_per_sample;
  uint32_t  samples_per_decode;
  float     bytes_per_sample;                 /* estimate, refreshed by every block header */
  uint8_t*  data; uint32_t data_cap, data_size;
  packet_t  packets[STREAM_MAX_PACKETS];
  uint32_t  write_pos, read_pos, collect_pos, free_packets;
  int32_t*  cache[SLAI_MAX_CHANNELS];         /* samples of the current block */
  uint32_t  cache_cap, cache_n, cache_pos;
  uint32_t  cur_block_size;
};

struct SLAStreamingDecoder* SLAStreamingDecoder_Create(const struct SLAStreamingDecoderConfig* config)
{
  struct SLAStreamingDecoder* s;
  uint32_t ch;
  if (config == NULL || config->decode_interval_hz <= 0.0f) { return NULL; }
  s = (struct SLAStreamingDecoder*)calloc(1, sizeof(*s));
  if (s == NULL) { return NULL; }
  s->core = SLADecoder_Create(&config->core_config);
  if (s->core == NULL) { free(s); return NULL; }
  s->decode_interval_hz = config->decode_interval_hz;
  s->max_bit_per_sample = config->max_bit_per_sample;
  s->data_cap = 2u * SLA_CalculateSufficientBlockSize(config->core_config.max_num_channels, config->core_config.max_num_block_samples,
                                                      config->max_bit_per_sample);
  if (s->data_cap < 64) { s->data_cap = 64; }
  s->data = (uint8_t*)calloc(s->data_cap, 1);
  s->cache_cap = config->core_config.max_num_block_samples;
  for (ch = 0; ch < config->core_config.max_num_channels; ch++) { s->cache[ch] = (int32_t*)malloc(sizeof(int32_t) * (s->cache_cap + 1)); }
  s->bytes_per_sample = (float)((double)config->core_config.max_num_channels * (config->max_bit_per_sample / 8));
  s->free_packets = STREAM_MAX_PACKETS;
  if (s->data == NULL) { SLAStreamingDecoder_Destroy(s); return NULL; }
  for (ch = 0; ch < config->core_config.max_num_channels; ch++) { if (s->cache[ch] == NULL) { SLAStreamingDecoder_Destroy(s); return NULL; } }
  return s;
}

void SLAStreamingDecoder_Destroy(struct SLAStreamingDecoder* s)
{
  uint32_t ch;
  if (s == NULL) { return; }
  SLADecoder_Destroy(s->core);
  for (ch = 0; ch < SLAI_MAX_CHANNELS; ch++) { free(s->cache[ch]); }
  free(s->data);
  free(s);
}

SLAApiResult SLAStreamingDecoder_SetWaveFormat(struct SLAStreamingDecoder* s, const struct SLAWaveFormat* wave_format)
{
  SLAApiResult ret;
  if (s == NULL) { return SLA_APIRESULT_INVALID_ARGUMENT; }
  if ((ret = SLADecoder_SetWaveFormat(s->core, wave_format)) != SLA_APIRESULT_OK) { return ret; }
  if (wave_format->bit_per_sample > s->max_bit_per_sample) { return SLA_APIRESULT_EXCEED_HANDLE_CAPACITY; }
  s->samples_per_decode = (uint32_t)ceil(STREAM_MARGIN * (float)wave_format->sampling_rate / s->decode_interval_hz);
  return SLA_APIRESULT_OK;
}

SLAApiResult SLAStreamingDecoder_SetEncodeParameter(struct SLAStreamingDecoder* s, const struct SLAEncodeParameter* encode_param)
{
  if (s == NULL) { return SLA_APIRESULT_INVALID_ARGUMENT; }
  return SLADecoder_SetEncodeParameter(s->core, encode_param);
}

SLAApiResult SLAStreamingDecoder_EstimateMinimumNessesaryDataSize(struct SLAStreamingDecoder* s, uint32_t* estimate_data_size)
{
  uint32_t v;
  if (s == NULL || estimate_data_size == NULL) { return SLA_APIRESULT_INVALID_ARGUMENT; }
  v = (uint32_t)ceil((double)s->bytes_per_sample * s->samples_per_decode);
  *estimate_data_size = (v > DEC_MIN_BLOCK_HEADER) ? v : DEC_MIN_BLOCK_HEADER;
  return SLA_APIRESULT_OK;
}

static uint32_t queue_remain(const struct SLAStreamingDecoder* s)
{
  uint32_t pos, size = 0;
  if (s->free_packets == STREAM_MAX_PACKETS) { return 0; }
  pos = s->read_pos;
  do {
    size += s->packets[pos].size - s->packets[pos].used;
    pos = (pos + 1) % STREAM_MAX_PACKETS;
  } while (pos != s->write_pos);
  return size;
}

SLAApiResult SLAStreamingDecoder_GetRemainDataSize(struct SLAStreamingDecoder* s, uint32_t* remain_data_size)
{
  uint32_t inside = 0;
  if (s == NULL || remain_data_size == NULL) { return SLA_APIRESULT_INVALID_ARGUMENT; }
  /* the reference consumes a block's bytes as it hands its samples out; here the block is decoded at once, so the
   * share of its bytes that belongs to the samples still waiting is counted as remaining */
  if (s->cache_n > s->cache_pos && s->cache_n > 0) {
    inside = (uint32_t)(((uint64_t)s->cur_block_size * (s->cache_n - s->cache_pos)) / s->cache_n);
  }
  *remain_data_size = queue_remain(s) + s->data_size + inside;
  return SLA_APIRESULT_OK;
}

SLAApiResult SLAStreamingDecoder_EstimateDecodableNumSamples(struct SLAStreamingDecoder* s, uint32_t* estimate_num_samples)
{
  uint32_t remain;
  SLAApiResult ret;
  if (s == NULL || estimate_num_samples == NULL) { return SLA_APIRESULT_INVALID_ARGUMENT; }
  if ((ret = SLAStreamingDecoder_GetRemainDataSize(s, &remain)) != SLA_APIRESULT_OK) { return ret; }
  *estimate_num_samples = (uint32_t)floor((float)remain / s->bytes_per_sample);
  return SLA_APIRESULT_OK;
}

SLAApiResult SLAStreamingDecoder_GetOutputNumSamplesPerDecode(struct SLAStreamingDecoder* s, uint32_t* output_num_samples)
{
  if (s == NULL || output_num_samples == NULL) { return SLA_APIRESULT_INVALID_ARGUMENT; }
  *output_num_samples = s->samples_per_decode;
  return SLA_APIRESULT_OK;
}

/* queued fragments -> block buffer, as far as there is room */
static void stream_pull(struct SLAStreamingDecoder* s)
{
  while (s->free_packets != STREAM_MAX_PACKETS && s->data_size < s->data_cap) {
    packet_t* p = &s->packets[s->read_pos];
    uint32_t take;
    if (s->read_pos == s->write_pos && p->size == p->used) { break; }
    take = p->size - p->used;
    if (take > s->data_cap - s->data_size) { take = s->data_cap - s->data_size; }
    memcpy(s->data + s->data_size, p->data + p->used, take);
    s->data_size += take; p->used += take;
    if (p->used == p->size) { s->read_pos = (s->read_pos + 1) % STREAM_MAX_PACKETS; }
    if (take == 0) { break; }
  }
}

SLAApiResult SLAStreamingDecoder_AppendDataFragment(struct SLAStreamingDecoder* s, const uint8_t* data, uint32_t data_size)
{
  if (s == NULL || data == NULL) { return SLA_APIRESULT_INVALID_ARGUMENT; }
  if (s->free_packets == 0) { return SLA_APIRESULT_EXCEED_HANDLE_CAPACITY; }
  if (data_size != 0) {
    packet_t* p = &s->packets[s->write_pos];
    p->data = data; p->size = data_size; p->used = 0;
    s->write_pos = (s->write_pos + 1) % STREAM_MAX_PACKETS;
    s->free_packets--;
  }
  stream_pull(s);
  return SLA_APIRESULT_OK;
}

SLAApiResult SLAStreamingDecoder_CollectDataFragment(struct SLAStreamingDecoder* s, const uint8_t** data_ptr, uint32_t* data_size)
{
  packet_t* p;
  if (s == NULL || data_ptr == NULL || data_size == NULL) { return SLA_APIRESULT_INVALID_ARGUMENT; }
  if (s->free_packets == STREAM_MAX_PACKETS) { return SLA_APIRESULT_NO_DATA_FRAGMENTS; }
  p = &s->packets[s->collect_pos];
  if (p->used == 0) { return SLA_APIRESULT_NO_DATA_FRAGMENTS; }
  *data_ptr = p->data; *data_size = p->used;
  p->size -= p->used; p->data += p->used; p->used = 0;
  if (p->size == 0) { s->collect_pos = (s->collect_pos + 1) % STREAM_MAX_PACKETS; s->free_packets++; }
  return SLA_APIRESULT_OK;
}

SLAApiResult SLAStreamingDecoder_Decode(struct SLAStreamingDecoder* s, int32_t** buffer, uint32_t buffer_num_samples,
                                        uint32_t* num_output_samples)
{
  uint32_t goal, progress = 0, ch;
  const uint32_t C = (s != NULL) ? s->core->wave_format.num_channels : 0;
  if (s == NULL || buffer == NULL || num_output_samples == NULL) { return SLA_APIRESULT_INVALID_ARGUMENT; }
  if (!(s->core->status_flag & STATUS_WAVE_FORMAT) || !(s->core->status_flag & STATUS_ENCODE_PARAM)) { return SLA_APIRESULT_PARAMETER_NOT_SET; }
  goal = (buffer_num_samples < s->samples_per_decode) ? buffer_num_samples : s->samples_per_decode;
  while (progress < goal) {
    uint32_t take;
    if (s->cache_pos >= s->cache_n) {
      /* next block: header fields first (src/SLADecoder.c:1031-1049), then the whole block once it is here */
      uint32_t bsize, nsmpl = 0, used = 0;
      SLAApiResult ret;
      stream_pull(s);
      if (s->data_size < DEC_MIN_BLOCK_HEADER) {
        if (progress > 0) { break; }
        return SLA_APIRESULT_INSUFFICIENT_DATA_SIZE;
      }
      if (rd_be16(s->data) != SLAI_SYNC_CODE) { return SLA_APIRESULT_FAILED_TO_FIND_SYNC_CODE; }
      bsize = rd_be32(s->data + 2) + 6u;
      if (bsize > s->data_size) { break; }                      /* the rest of the block has not been appended yet */
      ret = decode_one_block(s->core, s->data, s->data_size, s->cache, s->cache_cap, &used, &nsmpl);
      if (ret != SLA_APIRESULT_OK) { return ret; }
      if (used == 0 || used > s->data_size) { return SLA_APIRESULT_NG; }
      s->bytes_per_sample = (nsmpl > 0) ? (float)((double)bsize / nsmpl) : s->bytes_per_sample;
      s->cur_block_size = bsize;
      memmove(s->data, s->data + used, s->data_size - used);      /* src/SLADecoder.c:1086-1092 */
      s->data_size -= used;
      s->cache_n = nsmpl; s->cache_pos = 0;
      stream_pull(s);
      if (nsmpl == 0) { continue; }
    }
    take = s->cache_n - s->cache_pos;
    if (take > goal - progress) { take = goal - progress; }
    for (ch = 0; ch < C; ch++) { memcpy(buffer[ch] + progress, s->cache[ch] + s->cache_pos, sizeof(int32_t) * take); }
    s->cache_pos += take; progress += take;
  }
  *num_output_samples = progress;
  return SLA_APIRESULT_OK;
}
