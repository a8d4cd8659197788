/*
 * sla_flat.h -- flat (ctypes-friendly) parameter and trace structures shared by
 * the CPU oracle (sla_oracle.c) and the reference probe (ref_probe.c).
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is part of the product
 * path: only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg
 * may load it.
 */
#ifndef SLA_FLAT_H_INCLUDED
#define SLA_FLAT_H_INCLUDED

#include <stdint.h>

/* Encoder set-up.  First block mirrors SLAWaveFormat + SLAEncodeParameter
 * (reference src/include/public/SLA.h:61-76), second block mirrors
 * SLAEncoderConfig (src/include/public/SLAEncoder.h:14-21): the capacity fixes
 * the long-term analyser's FFT size (src/SLAEncoder.c:110). */
typedef struct sla_flat_params {
  uint32_t num_channels;
  uint32_t bits_per_sample;
  uint32_t sampling_rate;
  uint32_t parcor_order;
  uint32_t longterm_order;
  uint32_t lms_order;
  uint32_t ch_process_method;   /* 0 none, 1 stereo MS            */
  uint32_t window_type;         /* 0 rect 1 sin 2 hann 3 blackman 4 vorbis */
  uint32_t max_block_samples;
  uint32_t cap_channels;
  uint32_t cap_block_samples;
  uint32_t cap_parcor_order;
  uint32_t cap_longterm_order;
  uint32_t cap_lms_order;
} sla_flat_params;

/* Per-block intermediates of a whole-file encode.  All arrays are caller
 * allocated; per-(block,channel) arrays are indexed (b * num_channels + ch). */
typedef struct sla_flat_trace {
  uint32_t  max_blocks;      /* in  */
  uint32_t  order_stride;    /* in: parcor_order + 1 */
  uint32_t  ltm_stride;      /* in: longterm_order   */
  uint32_t  sample_stride;   /* in: per-channel stride of res_* (>= num_samples) */
  uint32_t  num_blocks;      /* out */
  uint32_t  offset_lshift;   /* out */
  uint32_t* blk_start;       /* [max_blocks] first sample of block           */
  uint32_t* blk_nsmpl;       /* [max_blocks] samples per channel in block    */
  uint32_t* blk_type;        /* [max_blocks] 0 compressed, 1 silent, 2 raw   */
  uint32_t* blk_bytes;       /* [max_blocks] encoded size of block           */
  double*   parcor;          /* [max_blocks*C*order_stride] analysis PARCOR  */
  int32_t*  code;            /* [max_blocks*C*order_stride] transmitted code */
  int32_t*  kint;            /* [max_blocks*C*order_stride] lattice coef     */
  uint32_t* rshift;          /* [max_blocks*C]                               */
  uint32_t* pitch;           /* [max_blocks*C]                               */
  int32_t*  ltm_coef;        /* [max_blocks*C*ltm_stride] (<<16 form)        */
  uint32_t* rice_init;       /* [max_blocks*C]                               */
  int32_t*  res_lattice;     /* [C*sample_stride] residual after lattice     */
  int32_t*  res_final;       /* [C*sample_stride] residual after LTM + LMS   */
} sla_flat_trace;

#endif
