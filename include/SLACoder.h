/*
 * SLACoder.h -- the part of the reference's coder API (src/include/private/SLACoder.h:15-24) that lies on the
 * encode hot path: the initial recursive-Rice parameters of a block, implemented over the MI355X kernels of
 * libsla_hip.so (one call = one block; see SLAPredictor.h for what that means).
 *
 * Replaces (reference file:line)
 *   SLACoder_Create / _Destroy                                  src/SLACoder.c:321-358
 *   SLACoder_CalculateInitialRecursiveRiceParameter             src/SLACoder.c:361-385
 *
 * The reference keeps the parameters inside the handle and only its bit-stream writers read them back
 * (SLACoder_PutInitialRecursiveRiceParameter, :388-404).  The bit-stream side of this library is the device bit-pack
 * behind SLAEncoder_EncodeWhole (sla_hip_pack_device), so the handle exposes the computed values instead:
 * sla_hip_coder_initial_parameter().
 */
#ifndef SLACODER_H_INCLUDED
#define SLACODER_H_INCLUDED

#include <stdint.h>

struct SLACoder;

#ifdef __cplusplus
extern "C" {
#endif

struct SLACoder* SLACoder_Create(uint32_t max_num_channels, uint32_t max_num_parameters);
void SLACoder_Destroy(struct SLACoder* coder);

/* per channel: max(mean of the zig-zag folded residual, 1), stored for every one of the num_parameters stages */
void SLACoder_CalculateInitialRecursiveRiceParameter(
    struct SLACoder* coder, uint32_t num_parameters,
    const int32_t** data, uint32_t num_channels, uint32_t num_samples);

/* value computed by the last SLACoder_CalculateInitialRecursiveRiceParameter for `channel` (0 if out of range) */
uint32_t sla_hip_coder_initial_parameter(const struct SLACoder* coder, uint32_t channel);

#ifdef __cplusplus
}
#endif

#endif /* SLACODER_H_INCLUDED */
