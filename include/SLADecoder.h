/*
 * SLADecoder.h -- whole-file decoder entry points of libsla_hip.so (MI355X / gfx950).
 *
 * Same names, argument order, struct layout and result codes as the reference decoder API
 * (reference src/include/public/SLADecoder.h:16-60), so the reference CLI (src/main.c:156-216) links against
 * this library unchanged and gets sample-identical PCM.  The work behind SLADecoder_DecodeWhole is batched:
 * the host only walks the block chain (10 bytes per block), everything else -- CRC16, block header fields,
 * entropy decoding, LMS / long-term / PARCOR synthesis, de-emphasis, mid/side, left-justification -- runs
 * as HIP kernels over all blocks of the file at once (sla_hip.h, "decode side").
 *
 * Error behaviour follows the reference block by block: blocks are examined in file order, the first
 * failing one decides the result (src/SLADecoder.c:696-722), and the samples of the blocks before it have
 * been written to `buffer`; *output_num_samples then holds their count (the reference leaves it unwritten on
 * failure, src/SLADecoder.c:729).
 *
 * Differences a caller can observe:
 *   - SLADecoder_Create returns NULL when no HIP device is usable (there is no CPU fallback) and for
 *     capacities the kernels do not cover: more than 8 channels, blocks above 16384 samples (long-term
 *     synthesis keeps a block in LDS), more than 5 long-term taps; a stream whose LMS filters are not 4, 8,
 *     16 or 32 coefficients long fails with SLA_APIRESULT_FAILED_TO_SYNTHESIZE at its first compressed block;
 *   - a block whose header announces more samples than max_num_block_samples is refused with
 *     SLA_APIRESULT_INSUFFICIENT_BUFFER_SIZE (the reference writes past its work buffers);
 *   - a block type the format does not define returns SLA_APIRESULT_INVALID_HEADER_FORMAT (the reference
 *     asserts).
 * The streaming decoder (SLAStreamingDecoder_*, reference src/SLADecoder.c:735-1123) is here too, with the
 * reference's entry points and fragment bookkeeping; its unit of device work is a whole block (see below).
 */
#ifndef SLA_DECODER_H_INCLUDED
#define SLA_DECODER_H_INCLUDED

#include "SLA.h"

#define SLA_DECODER_VERSION_STRING   "0.0.1(beta)"

struct SLADecoder;                     /* opaque: device buffers, stream, host block table */
struct SLAStreamingDecoder;            /* opaque: a decoder + fragment queue + block buffer       */

/* Capacity of a handle (layout = reference src/include/public/SLADecoder.h:16-24). */
struct SLADecoderConfig {
  uint32_t max_num_channels;           /* 1..8 planes                                            */
  uint32_t max_num_block_samples;      /* <= 16384 here                                          */
  uint32_t max_parcor_order;           /* PARCOR coefficients per channel, <= 255                */
  uint32_t max_longterm_order;         /* long-term taps, <= 5                                   */
  uint32_t max_lms_order_per_filter;   /* LMS coefficients per cascade stage                     */
  uint8_t  enable_crc_check;           /* 1: verify the CRC16 of every block (on the device)     */
  uint8_t  verpose_flag;               /* (sic) unused                                           */
};

/* layout = reference src/include/public/SLADecoder.h:27-31 */
struct SLAStreamingDecoderConfig {
  struct SLADecoderConfig core_config;
  float                   decode_interval_hz;   /* Decode calls per second of audio          */
  uint32_t                max_bit_per_sample;   /* sizes the block buffer                    */
};

#ifdef __cplusplus
extern "C" {
#endif

/* The 43-byte file header, host only.  A CRC mismatch still fills *header_info and returns
 * SLA_APIRESULT_DETECT_DATA_CORRUPTION.  reference src/SLADecoder.c:171-251 */
SLAApiResult
SLADecoder_DecodeHeader(const uint8_t*        data,
                        uint32_t              data_size,
                        struct SLAHeaderInfo* header_info);

/* Life cycle.  reference src/SLADecoder.c:69-128 and :131-168 */
struct SLADecoder*
SLADecoder_Create(const struct SLADecoderConfig* config);

void
SLADecoder_Destroy(struct SLADecoder* decoder);

/* Stream format and coding parameters (DecodeWhole sets both from the file header).
 * reference src/SLADecoder.c:254-275 and :278-302 */
SLAApiResult
SLADecoder_SetWaveFormat(struct SLADecoder*          decoder,
                         const struct SLAWaveFormat* wave_format);

SLAApiResult
SLADecoder_SetEncodeParameter(struct SLADecoder*               decoder,
                              const struct SLAEncodeParameter* encode_param);

/* Header + every block of a .sla image -> planar PCM, left-justified in 32 bits, buffer[ch][n].
 * reference src/SLADecoder.c:660-732 (DecodeBlock :583-657, block header :305-412, body :417-566) */
SLAApiResult
SLADecoder_DecodeWhole(struct SLADecoder* decoder,
                       const uint8_t*     data,
                       uint32_t           data_size,
                       int32_t**          buffer,
                       uint32_t           buffer_num_samples,
                       uint32_t*          output_num_samples);

/* ---- streaming decoder (reference src/SLADecoder.c:735-1123) -------------------------------------------------
 * Append fragments of the stream (referenced in place, at most 8 outstanding; CollectDataFragment hands back what
 * has been copied into the block buffer), then draw ceil(1.05 * sampling_rate / decode_interval_hz) samples per
 * Decode call.  A block is decoded on the device as soon as all its bytes have arrived and calls are served from
 * its samples, so the PCM is exactly DecodeWhole's.  Difference: where the reference starts on a block whose tail
 * has not been appended yet, Decode returns the samples it has (possibly 0) with SLA_APIRESULT_OK and continues
 * once the rest is there; SLA_APIRESULT_INSUFFICIENT_DATA_SIZE means that not even a block header is available. */
struct SLAStreamingDecoder*
SLAStreamingDecoder_Create(const struct SLAStreamingDecoderConfig* config);

void
SLAStreamingDecoder_Destroy(struct SLAStreamingDecoder* decoder);

SLAApiResult
SLAStreamingDecoder_SetWaveFormat(struct SLAStreamingDecoder* decoder,
                                  const struct SLAWaveFormat* wave_format);

SLAApiResult
SLAStreamingDecoder_SetEncodeParameter(struct SLAStreamingDecoder*      decoder,
                                       const struct SLAEncodeParameter* encode_param);

/* bytes worth one Decode call at the current block's average rate (at least a block header)   :877-900 */
SLAApiResult
SLAStreamingDecoder_EstimateMinimumNessesaryDataSize(struct SLAStreamingDecoder* decoder,
                                                     uint32_t*                   estimate_data_size);

/* samples the appended-but-undecoded bytes are worth at that rate                                :903-929 */
SLAApiResult
SLAStreamingDecoder_EstimateDecodableNumSamples(struct SLAStreamingDecoder* decoder,
                                                uint32_t*                   estimate_num_samples);

SLAApiResult
SLAStreamingDecoder_GetOutputNumSamplesPerDecode(struct SLAStreamingDecoder* decoder,
                                                 uint32_t*                   output_num_samples);

SLAApiResult
SLAStreamingDecoder_AppendDataFragment(struct SLAStreamingDecoder* decoder,
                                       const uint8_t*              data,
                                       uint32_t                    data_size);

SLAApiResult
SLAStreamingDecoder_CollectDataFragment(struct SLAStreamingDecoder* decoder,
                                        const uint8_t**             data_ptr,
                                        uint32_t*                   data_size);

SLAApiResult
SLAStreamingDecoder_GetRemainDataSize(struct SLAStreamingDecoder* decoder,
                                      uint32_t*                   remain_data_size);

SLAApiResult
SLAStreamingDecoder_Decode(struct SLAStreamingDecoder* decoder,
                           int32_t**                   buffer,
                           uint32_t                    buffer_num_samples,
                           uint32_t*                   num_output_samples);

/* Wall time [ms] of the last DecodeWhole, 6 floats: upload, block walk, kernels (device, stream events),
 * download, total, number of kernel batches (1 unless a block's size field disagreed with its contents). */
int sla_hip_decoder_last_timing(const struct SLADecoder* decoder, float* timing_ms);

/* The same decode on an image that already lives in device memory, planes left on the device:
 * d_planes = [num_channels][plane_stride] int32 (plane_stride >= header.num_samples + 65535 or the caller's
 * capacity), no PCIe traffic except the 10-byte-per-block walk, which reads `host_data` (the same bytes). */
SLAApiResult sla_hip_decode_device(struct SLADecoder* decoder, const uint8_t* host_data, const uint32_t* d_image,
                                   uint32_t data_size, int32_t* d_planes, uint64_t plane_stride,
                                   uint32_t* output_num_samples);

#ifdef __cplusplus
}
#endif

#endif /* SLA_DECODER_H_INCLUDED */
