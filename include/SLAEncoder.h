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

struct SLAEncoder;

struct SLAEncoderConfig {
  uint32_t max_num_channels;
  uint32_t max_num_block_samples;
  uint32_t max_parcor_order;
  uint32_t max_longterm_order;
  uint32_t max_lms_order_per_filter;
  uint8_t  verpose_flag;               /* (sic) */
};

#ifdef __cplusplus
extern "C" {
#endif

/* replaces reference src/SLAEncoder.c:56-128 */
struct SLAEncoder* SLAEncoder_Create(const struct SLAEncoderConfig* config);
/* replaces reference src/SLAEncoder.c:131-173 */
void SLAEncoder_Destroy(struct SLAEncoder* encoder);
/* replaces reference src/SLAEncoder.c:176-197 */
SLAApiResult SLAEncoder_SetWaveFormat(struct SLAEncoder* encoder, const struct SLAWaveFormat* wave_format);
/* replaces reference src/SLAEncoder.c:200-224 */
SLAApiResult SLAEncoder_SetEncodeParameter(struct SLAEncoder* encoder, const struct SLAEncodeParameter* encode_param);
/* replaces reference src/SLAEncoder.c:227-292 */
SLAApiResult SLAEncoder_EncodeHeader(const struct SLAHeaderInfo* header, uint8_t* data, uint32_t data_size);
/* replaces reference src/SLAEncoder.c:458-801 (one block; host PCM in, bytes out) */
SLAApiResult SLAEncoder_EncodeBlock(struct SLAEncoder* encoder,
    const int32_t* const* input, uint32_t num_samples,
    uint8_t* data, uint32_t data_size, uint32_t* output_size);
/* replaces reference src/SLAEncoder.c:804-932 (header + every block of a file) */
SLAApiResult SLAEncoder_EncodeWhole(struct SLAEncoder* encoder,
    const int32_t* const* input, uint32_t num_samples,
    uint8_t* data, uint32_t data_size, uint32_t* output_size);

#ifdef __cplusplus
}
#endif

#endif /* SLA_ENCODER_H_INCLUDED */
