/*
 * SLAEncoder.h -- encoder entry points of libsla_hip.so (MI355X / gfx950).
 *
 * Signature-for-signature replacement of the reference encoder API
 * (reference src/include/public/SLAEncoder.h:14-53): a caller such as the
 * reference CLI (src/main.c:94-153) links against this library instead of
 * libsla.a and gets byte-identical .sla output, with the per-block
 * autocorrelation / Levinson-Durbin / PARCOR-lattice / Rice-parameter path
 * (reference src/SLAPredictor.c, src/SLACoder.c:361-385) running as HIP
 * kernels.  Error behaviour follows the reference: enum return codes, NULL
 * arguments -> SLA_APIRESULT_INVALID_ARGUMENT, capacity violations ->
 * SLA_APIRESULT_EXCEED_HANDLE_CAPACITY (reference src/SLAEncoder.c:176-224).
 *
 * There is no CPU fallback: SLAEncoder_Create returns NULL when no HIP
 * device is usable.
 */
#ifndef SLA_ENCODER_H_INCLUDED
#define SLA_ENCODER_H_INCLUDED

#include "SLA.h"

#define SLA_ENCODER_VERSION_STRING   "0.0.1(beta)"

struct SLAEncoder;                     /* opaque: device workspace, streams, host tables */

/* Capacity of a handle (layout = reference src/include/public/SLAEncoder.h:14-21). */
struct SLAEncoderConfig {
  uint32_t max_num_channels;           /* 1..8 planes                                                   */
  uint32_t max_num_block_samples;      /* <= 16384 here: the analysis window lives in LDS               */
  uint32_t max_parcor_order;           /* PARCOR coefficients per channel                               */
  uint32_t max_longterm_order;         /* long-term taps, 1 / 3 / 5                                     */
  uint32_t max_lms_order_per_filter;   /* LMS coefficients per cascade stage                            */
  uint8_t  verpose_flag;               /* (sic) unused by the encoder                                   */
};

#ifdef __cplusplus
extern "C" {
#endif

/* Life cycle.  Create: reference src/SLAEncoder.c:56-128 (here it also brings up the HIP device: streams,
 * events, twiddle tables, the host thread pool).  Destroy: :131-173. */
struct SLAEncoder*
SLAEncoder_Create(const struct SLAEncoderConfig* config);

void
SLAEncoder_Destroy(struct SLAEncoder* encoder);

/* Stream format and coding parameters; both must be set before any encode call.
 * reference src/SLAEncoder.c:176-197 and :200-224 */
SLAApiResult
SLAEncoder_SetWaveFormat(struct SLAEncoder*          encoder,
                         const struct SLAWaveFormat* wave_format);

SLAApiResult
SLAEncoder_SetEncodeParameter(struct SLAEncoder*               encoder,
                              const struct SLAEncodeParameter* encode_param);

/* The 43-byte file header, host only.  reference src/SLAEncoder.c:227-292 */
SLAApiResult
SLAEncoder_EncodeHeader(const struct SLAHeaderInfo* header,
                        uint8_t*                    data,
                        uint32_t                    data_size);

/* Everything of a file in one call -- header, partition search, every block: the batched device pipeline.
 * input[ch][n]: planar PCM, left-justified in 32 bits.  reference src/SLAEncoder.c:804-932 */
SLAApiResult
SLAEncoder_EncodeWhole(struct SLAEncoder*    encoder,
                       const int32_t* const* input,
                       uint32_t              num_samples,
                       uint8_t*              data,
                       uint32_t              data_size,
                       uint32_t*             output_size);

/* One block (host PCM in, block bytes out; a PCIe round trip per call).  reference src/SLAEncoder.c:458-801 */
SLAApiResult
SLAEncoder_EncodeBlock(struct SLAEncoder*    encoder,
                       const int32_t* const* input,
                       uint32_t              num_samples,
                       uint8_t*              data,
                       uint32_t              data_size,
                       uint32_t*             output_size);

#ifdef __cplusplus
}
#endif

#endif /* SLA_ENCODER_H_INCLUDED */
