/*
 * SLAPredictor.h -- the encode-side predictor API of the reference (src/include/private/SLAPredictor.h),
 * restated with identical names, argument order, meaning and result codes, and implemented over the MI355X
 * kernels of libsla_hip.so.  One call = one block (that is the reference's granularity): every
 * call uploads its operands, runs the batched kernel on a batch of one and copies the result back, so these
 * entry points are for drop-in compatibility and for testing the kernels one stage at a time -- throughput
 * comes from SLAEncoder_EncodeWhole / sla_hip_analyze_device, which run the same kernels over a whole file.
 *
 * Replaces (reference file:line)
 *   SLALPCCalculator_Create / _Destroy                          src/SLAPredictor.c:148-186
 *   SLALPCCalculator_CalculatePARCORCoefDouble                  src/SLAPredictor.c:189-214 (autocorrelation :331-388,
 *                                                               Levinson-Durbin :253-328)
 *   SLALPCCalculator_EstimateCodeLength                         src/SLAPredictor.c:416-468
 *   SLALPCSynthesizer_Create / _Destroy / _Reset                src/SLAPredictor.c:505-554
 *   SLALPCSynthesizer_PredictByParcorCoefInt32                  src/SLAPredictor.c:557-607
 *   SLALongTermCalculator_Create / _Destroy / _CalculateCoef    src/SLAPredictor.c:743-980
 *   SLALongTermSynthesizer_Create / _Destroy / _Reset /
 *     _PredictInt32                                             src/SLAPredictor.c:983-1119
 *   SLALMSFilter_Create / _Destroy / _Reset / _PredictInt32     src/SLAPredictor.c:1133-1331
 *   SLAOptimalEncodeEstimator_Create / _Destroy /
 *     _SearchOptimalBlockPartitions / _CalculateMaxNumPartitions src/SLAPredictor.c:1467-1518, 1584-1705
 *   SLAEmphasisFilter_Create / _Reset / _Destroy /
 *     _PreEmphasisInt32 / _PreEmphasisDouble                    src/SLAPredictor.c:1708-1765, 1794-1813
 *
 * The decode side is here as well, one block per call through the synthesis kernels of sla_decode.hip:
 *   SLALPCSynthesizer_SynthesizeByParcorCoefInt32               src/SLAPredictor.c:610-740
 *   SLALongTermSynthesizer_SynthesizeInt32                      src/SLAPredictor.c:1034-1119
 *   SLALMSFilter_SynthesizeInt32                                src/SLAPredictor.c:1334-1463
 *   SLAEmphasisFilter_DeEmphasisInt32                           src/SLAPredictor.c:1768-1791
 * Not provided: SLALPCCalculator_CalculateResidualPower, which neither the encoder nor the decoder calls.
 *
 * Differences a caller can observe:
 *   - every Create returns NULL without a HIP device (no CPU fallback);
 *   - the filters with memory (lattice, long-term, LMS) run a call from the reset state: call _Reset before each
 *     block, as src/SLAEncoder.c:594-659 does.  A second Predict on a handle that was not reset returns
 *     SLAPREDICTOR_APIRESULT_NG instead of continuing the previous block.  (The emphasis filter carries its
 *     previous sample across calls like the reference.)
 *   - limits: num_samples <= 16384 per call (analysis window in LDS), LMS coefficients 4/8/16/32, taps 1/3/5.
 */
#ifndef SLAPREDICTOR_H_INCLUDED
#define SLAPREDICTOR_H_INCLUDED

#include <stdint.h>

struct SLALPCCalculator;
struct SLALPCSynthesizer;
struct SLALongTermCalculator;
struct SLALongTermSynthesizer;
struct SLALMSFilter;
struct SLAOptimalBlockPartitionEstimator;
struct SLAEmphasisFilter;

/* result codes (values as in the reference header, src/include/private/SLAPredictor.h:28-34) */
typedef enum SLAPredictorApiResultTag {
  SLAPREDICTOR_APIRESULT_OK,
  SLAPREDICTOR_APIRESULT_NG,
  SLAPREDICTOR_APIRESULT_INVALID_ARGUMENT,
  SLAPREDICTOR_APIRESULT_EXCEED_MAX_ORDER,
  SLAPREDICTOR_APIRESULT_FAILED_TO_CALCULATION
} SLAPredictorApiResult;

#ifdef __cplusplus
extern "C" {
#endif

/* PARCOR coefficients of one block of (already windowed / pre-emphasised) doubles; parcor_coef has order+1 entries */
struct SLALPCCalculator* SLALPCCalculator_Create(uint32_t max_order);
void SLALPCCalculator_Destroy(struct SLALPCCalculator* lpc);
SLAPredictorApiResult SLALPCCalculator_CalculatePARCORCoefDouble(
    struct SLALPCCalculator* lpcc,
    const double* data, uint32_t num_samples,
    double* parcor_coef, uint32_t order);

/* estimated code length [bytes per sample] from the block's energy and its PARCOR coefficients */
SLAPredictorApiResult SLALPCCalculator_EstimateCodeLength(
    const double* data, uint32_t num_samples, uint32_t bits_per_sample,
    const double* parcor_coef, uint32_t order,
    double* length_per_sample);

/* PARCOR lattice: int32 samples -> int32 residual; parcor_coef has order+1 entries (16-bit fixed point) */
struct SLALPCSynthesizer* SLALPCSynthesizer_Create(uint32_t max_order);
void SLALPCSynthesizer_Destroy(struct SLALPCSynthesizer* lpc);
SLAPredictorApiResult SLALPCSynthesizer_Reset(struct SLALPCSynthesizer* lpc);
SLAPredictorApiResult SLALPCSynthesizer_PredictByParcorCoefInt32(
    struct SLALPCSynthesizer* lpcs,
    const int32_t* data, uint32_t num_samples,
    const int32_t* parcor_coef, uint32_t order,
    int32_t* residual);
/* the inverse: residual -> samples (IIR lattice) */
SLAPredictorApiResult SLALPCSynthesizer_SynthesizeByParcorCoefInt32(
    struct SLALPCSynthesizer* lpcs,
    const int32_t* residual, uint32_t num_samples,
    const int32_t* parcor_coef, uint32_t order,
    int32_t* output);

/* long-term (pitch) analysis: FFT autocorrelation, pitch pick, Toeplitz solve */
struct SLALongTermCalculator* SLALongTermCalculator_Create(
    uint32_t fft_size, uint32_t max_pitch_period,
    uint32_t max_num_pitch_candidates, uint32_t max_num_taps);
void SLALongTermCalculator_Destroy(struct SLALongTermCalculator* ltm_calculator);
SLAPredictorApiResult SLALongTermCalculator_CalculateCoef(
    struct SLALongTermCalculator* ltm_calculator,
    const int32_t* data, uint32_t num_samples,
    uint32_t* pitch_num_samples, double* ltm_coef, uint32_t num_taps);

/* long-term prediction: residual[n] = data[n] - sum_k coef[k] * data[n - pitch - taps/2 + k]  (Q31 taps) */
struct SLALongTermSynthesizer* SLALongTermSynthesizer_Create(uint32_t max_num_taps, uint32_t max_pitch_period);
void SLALongTermSynthesizer_Destroy(struct SLALongTermSynthesizer* ltm);
SLAPredictorApiResult SLALongTermSynthesizer_Reset(struct SLALongTermSynthesizer* ltm);
SLAPredictorApiResult SLALongTermSynthesizer_PredictInt32(
    struct SLALongTermSynthesizer* ltm,
    const int32_t* data, uint32_t num_samples,
    uint32_t pitch_period,
    const int32_t* ltm_coef, uint32_t num_taps, int32_t* residual);
/* the inverse: output[n] = residual[n] + sum_k coef[k] * output[n - pitch - taps/2 + k] */
SLAPredictorApiResult SLALongTermSynthesizer_SynthesizeInt32(
    struct SLALongTermSynthesizer* ltm,
    const int32_t* residual, uint32_t num_samples,
    uint32_t pitch_period,
    const int32_t* ltm_coef, uint32_t num_taps, int32_t* output);

/* sign-log LMS cascade */
struct SLALMSFilter* SLALMSFilter_Create(uint32_t max_num_coef);
void SLALMSFilter_Destroy(struct SLALMSFilter* nlms);
SLAPredictorApiResult SLALMSFilter_Reset(struct SLALMSFilter* nlms);
SLAPredictorApiResult SLALMSFilter_PredictInt32(
    struct SLALMSFilter* nlms, uint32_t num_coef,
    const int32_t* data, uint32_t num_samples, int32_t* residual);
SLAPredictorApiResult SLALMSFilter_SynthesizeInt32(
    struct SLALMSFilter* nlms, uint32_t num_coef,
    const int32_t* residual, uint32_t num_samples, int32_t* output);

/* block partition search over one super-frame (data[ch] = num_samples un-windowed doubles) */
struct SLAOptimalBlockPartitionEstimator* SLAOptimalEncodeEstimator_Create(
    uint32_t max_num_samples, uint32_t delta_num_samples);
void SLAOptimalEncodeEstimator_Destroy(struct SLAOptimalBlockPartitionEstimator* oee);
SLAPredictorApiResult SLAOptimalEncodeEstimator_SearchOptimalBlockPartitions(
    struct SLAOptimalBlockPartitionEstimator* oee,
    struct SLALPCCalculator* lpcc,
    const double* const* data, uint32_t num_channels, uint32_t num_samples,
    uint32_t min_num_block_samples, uint32_t delta_num_samples, uint32_t max_num_block_samples,
    uint32_t bits_per_sample, uint32_t parcor_order,
    uint32_t* optimal_num_partitions, uint32_t* optimal_block_partition);
uint32_t SLAOptimalEncodeEstimator_CalculateMaxNumPartitions(
    uint32_t max_num_samples, uint32_t delta_num_samples);

/* pre-emphasis y[n] = x[n] - ((x[n-1] * (2^s - 1)) >> s), in place */
struct SLAEmphasisFilter* SLAEmphasisFilter_Create(void);
SLAPredictorApiResult SLAEmphasisFilter_Reset(struct SLAEmphasisFilter* emp);
void SLAEmphasisFilter_Destroy(struct SLAEmphasisFilter* emp);
SLAPredictorApiResult SLAEmphasisFilter_PreEmphasisInt32(
    struct SLAEmphasisFilter* emp,
    int32_t* data, uint32_t num_samples, int32_t coef_shift);
void SLAEmphasisFilter_PreEmphasisDouble(double* data, uint32_t num_samples, int32_t coef_shift);
/* de-emphasis y[n] = x[n] + ((y[n-1] * (2^s - 1)) >> s), in place; y[-1] = last output of the previous call */
SLAPredictorApiResult SLAEmphasisFilter_DeEmphasisInt32(
    struct SLAEmphasisFilter* emp,
    int32_t* data, uint32_t num_samples, int32_t coef_shift);

#ifdef __cplusplus
}
#endif

#endif /* SLAPREDICTOR_H_INCLUDED */
