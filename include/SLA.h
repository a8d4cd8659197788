/*
 * SLA.h -- public types of the SLA codec API, as exposed by libsla_hip.so.
 *
 * Layout-compatible re-statement of the reference's public header
 * (reference src/include/public/SLA.h:1-88): same macro values, same enum
 * numbering, same struct field order, so a caller built against the
 * reference header links and runs against this library unchanged.
 */
#ifndef SLA_H_INCLUDED
#define SLA_H_INCLUDED

#include <stdint.h>

#define SLA_VERSION_STRING          "1.0.0"
#define SLA_FORMAT_VERSION          1
#define SLA_HEADER_SIZE             43          /* bytes, big-endian fields            */
#define SLA_BLOCK_HEADER_SIZE       10
#define SLA_NUM_SAMPLES_INVALID     0xFFFFFFFF
#define SLA_NUM_BLOCKS_INVALID      0xFFFFFFFF
#define SLA_MAX_BLOCK_SIZE_INVAILD  0xFFFFFFFF  /* (sic) spelling kept for source compatibility */

/* output bound for one block / one file (reference SLA.h:22-23) */
#define SLA_CalculateSufficientBlockSize(num_channels, num_samples, bit_per_sample) \
  (2 * (num_channels) * (num_samples) * ((bit_per_sample) / 8))

typedef enum SLAApiResultTag {
  SLA_APIRESULT_OK = 0,
  SLA_APIRESULT_NG,
  SLA_APIRESULT_INVALID_ARGUMENT,
  SLA_APIRESULT_EXCEED_HANDLE_CAPACITY,
  SLA_APIRESULT_INSUFFICIENT_BUFFER_SIZE,
  SLA_APIRESULT_INVAILD_CHPROCESSMETHOD,
  SLA_APIRESULT_FAILED_TO_CALCULATE_COEF,
  SLA_APIRESULT_FAILED_TO_PREDICT,
  SLA_APIRESULT_FAILED_TO_SYNTHESIZE,
  SLA_APIRESULT_INSUFFICIENT_DATA_SIZE,
  SLA_APIRESULT_INVALID_HEADER_FORMAT,
  SLA_APIRESULT_DETECT_DATA_CORRUPTION,
  SLA_APIRESULT_FAILED_TO_FIND_SYNC_CODE,
  SLA_APIRESULT_INVALID_WINDOWFUNCTION_TYPE,
  SLA_APIRESULT_NO_DATA_FRAGMENTS,
  SLA_APIRESULT_PARAMETER_NOT_SET
} SLAApiResult;

typedef enum SLAChannelProcessMethodTag {
  SLA_CHPROCESSMETHOD_NONE = 0,
  SLA_CHPROCESSMETHOD_STEREO_MS
} SLAChannelProcessMethod;

typedef enum SLAWindowFunctionTypeTag {
  SLA_WINDOWFUNCTIONTYPE_RECTANGULAR = 0,
  SLA_WINDOWFUNCTIONTYPE_SIN,
  SLA_WINDOWFUNCTIONTYPE_HANN,
  SLA_WINDOWFUNCTIONTYPE_BLACKMAN,
  SLA_WINDOWFUNCTIONTYPE_VORBIS
} SLAWindowFunctionType;

struct SLAWaveFormat {
  uint32_t num_channels;
  uint32_t bit_per_sample;
  uint32_t sampling_rate;
  uint8_t  offset_lshift;      /* zero low bits common to every sample */
};

struct SLAEncodeParameter {
  uint32_t                parcor_order;
  uint32_t                longterm_order;
  uint32_t                lms_order_per_filter;
  SLAChannelProcessMethod ch_process_method;
  SLAWindowFunctionType   window_function_type;
  uint32_t                max_num_block_samples;
};

struct SLAHeaderInfo {
  struct SLAWaveFormat      wave_format;
  struct SLAEncodeParameter encode_param;
  uint32_t                  num_samples;
  uint32_t                  num_blocks;
  uint32_t                  max_block_size;
  uint32_t                  max_bit_per_second;
};

#endif /* SLA_H_INCLUDED */
