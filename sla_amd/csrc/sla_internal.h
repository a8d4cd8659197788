/*
 * sla_internal.h -- declarations shared by the C host side of libsla_hip.so.
 * Host code is plain C (the reference is C89/C99); device code lives in
 * sla_kernels.hip behind the launchers of include/sla_hip.h.
 */
#ifndef SLA_INTERNAL_H_INCLUDED
#define SLA_INTERNAL_H_INCLUDED

#ifndef __HIP_PLATFORM_AMD__
#define __HIP_PLATFORM_AMD__ 1
#endif
#include <hip/hip_runtime_api.h>

#include <stddef.h>
#include <stdint.h>

#include "SLA.h"
#include "SLAEncoder.h"
#include "sla_hip.h"

/* format constants (reference src/include/private/SLAInternal.h:6-38) */
#define SLAI_MAX_CHANNELS         8
#define SLAI_SYNC_CODE            0xFFFFu
#define SLAI_LTM_MAX_PERIOD       256u
#define SLAI_LTM_MIN_PITCH        3u
#define SLAI_LTM_PERIOD_BITS      10u
#define SLAI_MIN_BLOCK            2048u
#define SLAI_SEARCH_DELTA         1024u
#define SLAI_RICE_PARAMS          2u
#define SLAI_RICE_LOW_THRESHOLD   8u
#define SLAI_QUOT_THRESHOLD       16u
#define SLAI_PATH_PENALTY         300.0
#define SLAI_EST_BLOCK_HEADER     50.0
#define SLAI_RAW_THRESHOLD        0.95f
#define SLAI_BIG_WEIGHT           ((double)(1UL << 24))
#define SLAI_HDR_CRC_START        10
#define SLAI_BLK_CRC_START        8
#define SLAI_MAX_ORDER            255
#define SLAI_MAX_TAPS             5
#define SLAI_STREAM_LANES         6       /* worker lanes of a streamed SLAEncoder_EncodeWhole, at most */
#define SLAI_MAX_NODES            66      /* 65535 / 1024 + 2 */
enum { SLAI_BLK_COMPRESS = 0, SLAI_BLK_SILENT = 1, SLAI_BLK_RAW = 2 };

/* ---- sla_plan.c: host-side decisions fed by device results ---------------- */
int      slai_make_window(SLAWindowFunctionType type, double* w, uint32_t n);
double   slai_code_length(double sumsq, uint32_t n, uint32_t bps, const double* parcor, uint32_t order);
int      slai_shortest_path(const double* adj, uint32_t nodes, uint32_t* path);
int      slai_host_check(void);       /* 0: long double / double arithmetic of this host is the reference build's */
uint32_t slai_zero_run(const uint64_t* nz_mask, uint64_t from, uint64_t limit);
int      slai_range_is_zero(const uint64_t* nz_mask, uint64_t from, uint64_t count);

/* ---- sla_ltm.c: long-term predictor analysis ------------------------------ */
typedef struct slai_fft_plan slai_fft_plan;
slai_fft_plan* slai_fft_plan_create(uint32_t fft_size);
void           slai_fft_plan_destroy(slai_fft_plan* plan);
uint32_t       slai_fft_plan_size(const slai_fft_plan* plan);
void           slai_fft_plan_export(const slai_fft_plan* plan, double* out /* 3*fft_size doubles */);
/* pitch + taps from the device's compact autocorrelation record; returns 0 ok, 4 analysis failed */
int  slai_ltm_solve(const double* rec, uint32_t ntaps, uint32_t* pitch, double* coef);
#define SLAI_LTM_ACF_HEAD SLA_HIP_ACF_RECORD

/* ---- sla_pack.c: bit-serial block writer ---------------------------------- */
typedef struct slai_block_params {
  uint32_t num_samples;
  uint32_t type;
  uint32_t num_channels, order, ntaps, bps, lshift, mid_side;
  const int32_t*  code;        /* [C][order+1] */
  const uint32_t* rshift;      /* [C] */
  const uint32_t* pitch;       /* [C] */
  const int32_t*  ltm_q;       /* [C][SLAI_MAX_TAPS] */
  const uint32_t* rice_init;   /* [C] */
  const int32_t*  res[SLAI_MAX_CHANNELS];   /* final residual (COMPRESS) or right-justified PCM (RAW) */
} slai_block_params;
/* returns bytes written, or 0 when `cap` is too small */
uint32_t slai_pack_block(const slai_block_params* bp, uint8_t* out, uint32_t cap);
uint32_t slai_pack_header(const slai_block_params* bp, uint8_t* out, uint32_t cap);
void     slai_coding_mode(const uint32_t* rice_init, uint32_t num_channels, uint32_t* golomb_m);
uint32_t slai_crc16(const uint8_t* data, size_t n);
int      slai_write_header(const struct SLAHeaderInfo* h, uint8_t* data, uint32_t data_size);

#endif
