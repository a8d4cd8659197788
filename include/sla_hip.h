/*
 * sla_hip.h -- batched C-ABI of the MI355X (gfx950) SLA encode hot path.
 *
 * Plain C: pointers, sizes, an opaque stream handle.  No torch / C++ types.
 * Two layers:
 *
 *  (1) kernel launchers  sla_hip_launch_*  -- one call = one HIP kernel over a
 *      batch of descriptors that already live in device memory.  These are
 *      what replaces the reference's per-block inner functions:
 *        prepass  : SLAEncoder_CalculateLeftShiftOffset  src/SLAEncoder.c:425-455
 *                   + silence test                       src/SLAEncoder.c:392-408, 520-528
 *        lpc      : LPC_CalculateAutoCorrelation         src/SLAPredictor.c:331-388
 *                   LPC_LevinsonDurbinRecursion          src/SLAPredictor.c:253-328
 *                   (A0 staging: src/SLAEncoder.c:505-515, 540-543; src/SLAUtility.c:370-412)
 *                   Sum x^2 of SLALPCCalculator_EstimateCodeLength src/SLAPredictor.c:432-436
 *                   coefficient quantiser                src/SLAEncoder.c:567-589
 *        lattice  : SLAEmphasisFilter_PreEmphasisInt32   src/SLAPredictor.c:1741-1765
 *                   SLALPCSynthesizer_PredictByParcorCoefInt32 src/SLAPredictor.c:557-607
 *        ltm_acf  : FFT autocorrelation of SLALongTermCalculator_CalculateCoef src/SLAPredictor.c:827-853
 *        tail     : SLALongTermSynthesizer_PredictInt32  src/SLAPredictor.c:1031-1119
 *                   SLALMSFilter_PredictInt32            src/SLAPredictor.c:1202-1331
 *                   SLACoder_CalculateInitialRecursiveRiceParameter src/SLACoder.c:361-385
 *
 *  (2) the whole-file driver behind SLAEncoder_EncodeWhole (SLAEncoder.h),
 *      also callable on PCM that is already resident in HBM
 *      (sla_hip_analyze_device) -- this is what bench.py times.
 *
 * All functions return 0 on success, a negative hipError_t-derived code on a
 * HIP failure, or a positive SLAApiResult for API-level errors.
 */
#ifndef SLA_HIP_H_INCLUDED
#define SLA_HIP_H_INCLUDED

#include <stdint.h>
#include "SLA.h"
#include "SLAEncoder.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef void* sla_hip_stream_t;          /* a hipStream_t, or NULL for the default stream */

/* ---- descriptors (device-resident arrays of these feed the kernels) ----- */

/* One LPC work-group: a window of one channel staged once in LDS, analysed
 * over `cand_count` sub-ranges ("candidates" of the partition search, or the
 * single whole range of a chosen block). */
typedef struct sla_hip_lpc_group {
  uint64_t pcm_off;        /* first sample of the window inside a channel plane          */
  uint32_t num_samples;    /* window length W                                              */
  uint32_t channel;        /* output channel index (after mid/side when MS is on)         */
  uint32_t win_off;        /* offset of the analysis window table in the window pool, or
                              SLA_HIP_NO_WINDOW: rectangular, no pre-emphasis (search)     */
  uint32_t int_shift;      /* right shift of the integer path: 32 - bps + offset_lshift   */
  uint32_t cand_first;     /* first entry in the candidate table                          */
  uint32_t cand_count;
  uint32_t slot_first;     /* output slot of candidate 0; candidate c writes slot_first+c */
  uint32_t pad_;
} sla_hip_lpc_group;
#define SLA_HIP_NO_WINDOW 0xFFFFFFFFu
/* LDS one LPC work-group may use: window doubles + (order+1) doubles per candidate.
 * 160 KiB per CU on gfx950, minus a small static allocation. */
#define SLA_HIP_LDS_BUDGET (160u * 1024u - 256u)

typedef struct sla_hip_lpc_cand {
  uint32_t start;          /* relative to the group's window */
  uint32_t len;
} sla_hip_lpc_cand;

/* One lattice wave: `count` output samples of one (block, channel). */
typedef struct sla_hip_lattice_chunk {
  uint64_t blk_off;        /* first sample of the block inside a channel plane */
  uint32_t blk_len;
  uint32_t chunk_start;    /* first output sample of this chunk, relative to the block */
  uint32_t count;
  uint32_t channel;
  uint32_t slot;           /* (block, channel) slot: coefficients live at kint[slot*(order+1)] */
  uint32_t int_shift;
} sla_hip_lattice_chunk;

/* One (block, channel) of the long-term analysis FFT. */
typedef struct sla_hip_acf_job {
  uint64_t blk_off;
  uint32_t blk_len;
  uint32_t channel;
} sla_hip_acf_job;

/* One (block, channel) of the serial tail (long-term filter + LMS + Rice sum). */
typedef struct sla_hip_tail_job {
  uint64_t blk_off;
  uint32_t blk_len;
  uint32_t channel;
  uint32_t pitch;          /* 0 = long-term stage bypassed */
  int32_t  ltm_coef[5];    /* Q31 taps (<<16 form), longterm_order of them used */
  uint32_t pad_[2];
} sla_hip_tail_job;

/* One (block, channel) of the Rice code-length pass. */
typedef struct sla_hip_rice_job {
  uint64_t blk_off;
  uint32_t blk_len;
  uint32_t channel;
  uint32_t rice_init;      /* initial parameter (mean of the folded residual, >= 1)            */
  uint32_t golomb_m;       /* != 0: block is in fixed-parameter Golomb mode with this modulus */
} sla_hip_rice_job;

/* One block of the output image. */
typedef struct sla_hip_pack_block {
  uint64_t blk_off;        /* first sample inside a channel plane                              */
  uint64_t out_off;        /* byte offset of the block inside the .sla image                  */
  uint32_t num_samples;
  uint32_t type;           /* 0 compressed, 1 silent, 2 raw                                    */
  uint32_t header_off;     /* offset of the pre-packed header bytes in the header pool         */
  uint32_t header_bytes;
  uint32_t out_bytes;      /* total encoded size of the block                                  */
  uint32_t raw_bits;       /* RAW blocks: bits per sample of channel 0 (bps - offset_lshift)   */
  uint32_t golomb_m[8];    /* per channel: 0 = adaptive recursive Rice, else Golomb modulus    */
} sla_hip_pack_block;

/* ---- (1) kernel launchers ----------------------------------------------- */

/* Tuning knobs of the launchers.  Nothing on the launch path reads the environment: an encoder handle owns one of
 * these (filled once in SLAEncoder_Create, changed through sla_hip_encoder_set_option) and names it to the
 * launchers of the calling host thread before it launches; a thread that never called sla_hip_use_tuning (or
 * passed NULL) gets the defaults = all zeros.  None of the knobs changes a result, only how the work is laid out --
 * the three certification margins ("plan_margin", "cert_safety", "block_cert_safety") may only be WIDENED beyond their
 * built-in values (1e-4, 64, 16; plan_margin / cert_safety also take 0 = built-in value / no certificate): smaller
 * values would weaken the certificates byte-identity rests on, and sla_hip_encoder_set_option refuses them (a
 * plan_margin below 1e-4 in a sla_hip_tuning handed to sla_hip_use_tuning is read as 0). */
typedef struct sla_hip_tuning {
  uint32_t lpc_pack;            /* windows per workgroup of k_lpc / k_lpc_blocks, 0 = automatic                      */
  uint32_t lpc_threads;         /* 256 or 512 threads per k_lpc workgroup, 0 = automatic                             */
  uint32_t lpc_blocks_chains;   /* 1: chosen blocks through k_lpc's serial chains instead of k_lpc_blocks            */
  uint32_t tail_waves;          /* waves per tail workgroup (1..4), 0 = automatic (4)                                */
  uint32_t lpc_tile;            /* steps per tile of k_lpc_blocks' wide packs: 24, or 0 / 48 = 48 where it fits         */
  uint32_t tail_taps;           /* k_tailk: taps of each history per lane, 1 / 2 / 4; 0 = by the number of jobs (default) */
  double   plan_margin;         /* certification margin of k_plan, 0 = 1e-4 (tests raise it to force the host plan)  */
  uint32_t rice_lanes;          /* Rice parameter walk: 1 = one lane per job (k_rice_k), 2 = the two-lane pipeline (k_rice_k2), 0 = by the number of jobs */
  uint32_t lattice_plain;       /* 1: every lattice stage in the wrapping four-instruction form (round 3); 0 = the shortest form each stage's
                                   operand bound allows (same results: tests run both) */
  uint32_t cert_audit;          /* N > 0: every N-th (block, channel) pair that sla_hip_launch_lpc_blocks_cert CERTIFIES is analysed by the
                                   exact kernels as well, which compare codes, lattice coefficients and the RAW side with what the certified
                                   run stored: d_cert_flag 4 = audited and equal, 5 = the certificate was wrong (the encoder fails the call) */
} sla_hip_tuning;
void sla_hip_use_tuning(const sla_hip_tuning* tuning);

/* Optional extras of ONE launch, for the launchers that have an `_x` twin (same arguments + `const sla_hip_launch_extra*`, NULL =
 * none; the plain name is the twin with NULL).  What the encoder's driver needs and a caller of a single launcher usually
 * does not: */
typedef struct sla_hip_launch_extra {
  unsigned long long* d_span;       /* two device words (zeroed by the caller): [0] = max(~start), [1] = max(end) of the launch's execution on
                                       the device's constant 100 MHz clock -- first workgroup in to last workgroup out, without the time the
                                       launch waits for resources behind kernels of other streams (sla_hip_last_kernel_ms) */
  const uint32_t* d_group_count;    /* sla_hip_launch_lpc_blocks_cert_x: num_groups is an upper bound, the kernels take the number of groups from
                                       the running numbers sla_hip_launch_expand keeps on the device (the launch can be queued before the host
                                       has seen the count) */
  uint32_t* clear_ptr[3];           /* sla_hip_launch_search_exact_x: up to three regions of device words that the first workgroup of its first */
  uint32_t  clear_words[3];         /* kernel zeroes on the way (instead of one fill kernel each in front of it) */
} sla_hip_launch_extra;
int sla_hip_launch_lpc_x(const int32_t* d_pcm, uint64_t plane_stride, uint32_t mid_side, uint32_t order,
                         const sla_hip_lpc_group* d_groups, uint32_t num_groups, uint32_t max_window, uint32_t max_cands_per_group,
                         const sla_hip_lpc_cand* d_cands, const double* d_window_pool,
                         double* d_out, int32_t* d_code, int32_t* d_kint, uint32_t* d_rshift,
                         sla_hip_stream_t stream, const sla_hip_launch_extra* extra);
int sla_hip_launch_lpc_blocks_x(const int32_t* d_pcm, uint64_t plane_stride, uint32_t mid_side, uint32_t order,
                                const sla_hip_lpc_group* d_groups, uint32_t num_groups, uint32_t max_window,
                                const sla_hip_lpc_cand* d_cands, const double* d_window_pool,
                                double* d_out, int32_t* d_code, int32_t* d_kint, uint32_t* d_rshift,
                                int32_t* d_lattice_residual, sla_hip_stream_t stream, const sla_hip_launch_extra* extra);
int sla_hip_launch_lpc_blocks_cert_x(const int32_t* d_pcm, uint64_t plane_stride, uint32_t mid_side, uint32_t order,
                                     const sla_hip_lpc_group* d_groups, uint32_t num_groups, uint32_t max_window,
                                     const sla_hip_lpc_cand* d_cands, const double* d_window_pool,
                                     double* d_out, int32_t* d_code, int32_t* d_kint, uint32_t* d_rshift,
                                     uint32_t* d_cert_flag, uint32_t* d_fallback_list, uint32_t* d_fallback_count,
                                     double safety, uint32_t bits_per_sample, sla_hip_stream_t stream, const sla_hip_launch_extra* extra);

/* d_or_mask[0] = OR of every input word, d_or_mask[1] = number of all-zero words of the mask; d_nz_mask: one
 * bit per sample "any channel non-zero after right-justify / mid-side", ceil(num_samples/64) words. */
int sla_hip_launch_prepass(const int32_t* d_pcm, uint64_t plane_stride, uint32_t num_channels,
                           uint32_t num_samples, uint32_t bits_per_sample, uint32_t mid_side,
                           uint32_t* d_or_mask, uint64_t* d_nz_mask, sla_hip_stream_t stream);
/* The same pass that also stores the OR of every 1024-sample tile (d_tile_or: ceil(num_samples / 4096) * 4 words):
 * a batch of files laid out back to back on 1024-sample boundaries gets its offset_lshift per file from them. */
#define SLA_HIP_PREPASS_TILE 1024u
int sla_hip_launch_prepass_tiles(const int32_t* d_pcm, uint64_t plane_stride, uint32_t num_channels,
                                 uint32_t num_samples, uint32_t bits_per_sample, uint32_t mid_side,
                                 uint32_t* d_or_mask, uint64_t* d_nz_mask, uint32_t* d_tile_or, sla_hip_stream_t stream);

/* Per file of a batch laid out back to back on SLA_HIP_PREPASS_TILE boundaries (sla_hip_analyze_batch_device), from the results
 * of sla_hip_launch_prepass_tiles: d_info[3 i] = OR of the file's words, [3 i + 1] = all-zero 64-sample mask words inside it,
 * [3 i + 2] = 1 when its last super-frame (fewer than 127 samples left of it) is all zero.  No zero word and no such tail
 * anywhere: no block of the batch can be SILENT (reference src/SLAEncoder.c:392-408), the mask never has to leave the device. */
int sla_hip_launch_batch_scan(const uint64_t* d_nz_mask, const uint32_t* d_tile_or, const uint32_t* d_file_start,
                              const uint32_t* d_file_len, uint32_t num_files, uint32_t max_block_samples,
                              uint32_t* d_info, sla_hip_stream_t stream);

/* Autocorrelation + Levinson-Durbin for every candidate of every group.
 * d_out: per slot (order+2) doubles = { r[0], parcor[0..order] }.
 * When d_code/d_kint/d_rshift are non-NULL (chosen blocks: one candidate per
 * group covering the whole window) the coefficient quantiser runs too. */
int sla_hip_launch_lpc(const int32_t* d_pcm, uint64_t plane_stride, uint32_t mid_side, uint32_t order,
                       const sla_hip_lpc_group* d_groups, uint32_t num_groups, uint32_t max_window,
                       uint32_t max_cands_per_group,
                       const sla_hip_lpc_cand* d_cands, const double* d_window_pool,
                       double* d_out, int32_t* d_code, int32_t* d_kint, uint32_t* d_rshift,
                       sla_hip_stream_t stream);

/* Chosen blocks in one launch: sla_hip_launch_lpc in its quantiser form (one candidate per group = the whole
 * block) followed, inside the same workgroups, by the PARCOR lattice of sla_hip_launch_lattice with the
 * coefficients just quantised -> d_lattice_residual (plane layout of d_pcm). */
int sla_hip_launch_lpc_blocks(const int32_t* d_pcm, uint64_t plane_stride, uint32_t mid_side, uint32_t order,
                              const sla_hip_lpc_group* d_groups, uint32_t num_groups, uint32_t max_window,
                              const sla_hip_lpc_cand* d_cands, const double* d_window_pool,
                              double* d_out, int32_t* d_code, int32_t* d_kint, uint32_t* d_rshift,
                              int32_t* d_lattice_residual, sla_hip_stream_t stream);

/* Chosen blocks, certified route (the default of the block stage; option "block_cert"): the autocorrelation of every
 * windowed block in ANY summation order (k_acf_blocks: FMAs, one wave per block and channel), the reference's
 * Levinson-Durbin recursion on it and a per-block certificate that the quantised PARCOR codes (reference
 * src/SLAEncoder.c:567-589) and the RAW decision (:553-565) cannot differ from the reference's although the doubles may
 * differ in their last bits; blocks that do not certify are appended to d_fallback_list (group indices, count in
 * *d_fallback_count) and redone by the exact chain kernels of sla_hip_launch_lpc before the call's work on `stream`
 * ends.  d_cert_flag[slot]: 0 = certified, 2 = exact (4 / 5: audited, see sla_hip_tuning.cert_audit).  safety >= 1 scales the
 * certificate's bound (the encoder: 16).  The bound is a FIRST-ORDER perturbation bound of the Levinson-Durbin recursion times
 * that factor, validated empirically -- not a proof; the audit is the standing check on it.
 * Orders above 52 are not covered (SLA_APIRESULT_EXCEED_HANDLE_CAPACITY): use sla_hip_launch_lpc. */
int sla_hip_launch_lpc_blocks_cert(const int32_t* d_pcm, uint64_t plane_stride, uint32_t mid_side, uint32_t order,
                                   const sla_hip_lpc_group* d_groups, uint32_t num_groups, uint32_t max_window,
                                   const sla_hip_lpc_cand* d_cands, const double* d_window_pool,
                                   double* d_out, int32_t* d_code, int32_t* d_kint, uint32_t* d_rshift,
                                   uint32_t* d_cert_flag, uint32_t* d_fallback_list, uint32_t* d_fallback_count,
                                   double safety, uint32_t bits_per_sample, sla_hip_stream_t stream);

/* sla_hip_launch_lpc restricted to the groups that sla_hip_launch_search_exact flagged (NaN in r[0] of the group's
 * first slot): everything else returns at once and keeps its result.  *d_rerun_counter (may be NULL) is
 * incremented by the number of groups that were analysed. */
int sla_hip_launch_lpc_rerun(const int32_t* d_pcm, uint64_t plane_stride, uint32_t mid_side, uint32_t order,
                             const sla_hip_lpc_group* d_groups, uint32_t num_groups, uint32_t max_window,
                             uint32_t max_cands_per_group, const sla_hip_lpc_cand* d_cands,
                             double* d_out, uint32_t* d_rerun_counter, sla_hip_stream_t stream);

/* Partition search without serial chains (reference src/SLAPredictor.c:1615-1649 + :331-388).
 * The search analyses the un-windowed samples: integers times a power of two.  As long as the energy of
 * a group's window stays below `exact_limit` (= 2^51 units^2, unit = the common power-of-two factor of
 * the samples) every product and every partial sum of the reference's autocorrelation is exactly
 * representable, so the order of summation is free: each wave sums one SLA_HIP_XTILE-sample tile per lag,
 * candidates are prefix differences of tile sums minus the pairs that straddle the candidate's end, and
 * the result is bit-identical to the serial order.  Groups: one per (super-frame, channel) listing ALL its
 * candidates; candidates must start on a tile boundary and end on one or at the end of the window.
 * d_tile_sums: num_groups * SLA_HIP_XTILES * 2 * sla_hip_search_exact_lags(order) doubles of scratch.
 * A group whose energy reaches the limit: with cert_safety <= 0 it gets NaN in r[0] of every candidate and the
 * caller reruns it through sla_hip_launch_lpc_rerun.  With cert_safety > 0 (the encoder passes 64) its candidates
 * keep their tile-sum results -- close to, but no longer bit-identical with, the reference's serially rounded sums --
 * in the slot layout { r0, w, log2(e_p / r0), 0, .. }: parcor[0] (0 for every exact candidate) holds the half width w
 * of log2(e_p), e_p = r0 * prod(1 - k_j^2), parcor[1] the logarithm itself instead of a coefficient (orders <= 64):
 * the reference's own value of log2(e_p) provably lies within +-w (Loewner-order bracket of the prediction error over
 * every Toeplitz matrix within cert_safety x the summation error bounds of both sides; see k_search_finish), or
 * +inf when no such bracket exists.  The code-length estimate depends on the autocorrelation only through e_p, so
 * sla_hip_launch_plan can decide whether the partition is certain. */
#define SLA_HIP_XTILE  1024u
#define SLA_HIP_XTILES 16u
uint32_t sla_hip_search_exact_lags(uint32_t order);      /* padded lag count, 0: order not supported */
int sla_hip_launch_search_exact(const int32_t* d_pcm, uint64_t plane_stride, uint32_t mid_side, uint32_t order,
                                const sla_hip_lpc_group* d_groups, uint32_t num_groups, uint32_t max_window,
                                uint32_t max_cands_per_group,
                                const sla_hip_lpc_cand* d_cands, double* d_tile_sums, double* d_out,
                                double exact_limit, double cert_safety, uint32_t* d_any_exact /* one scratch word, may be NULL */,
                                sla_hip_stream_t stream);
int sla_hip_launch_search_exact_x(const int32_t* d_pcm, uint64_t plane_stride, uint32_t mid_side, uint32_t order,
                                const sla_hip_lpc_group* d_groups, uint32_t num_groups, uint32_t max_window,
                                uint32_t max_cands_per_group,
                                const sla_hip_lpc_cand* d_cands, double* d_tile_sums, double* d_out,
                                double exact_limit, double cert_safety, uint32_t* d_any_exact /* one scratch word, may be NULL */,
                                sla_hip_stream_t stream, const sla_hip_launch_extra* extra);

/* The scalar tail of the partition search on the device: estimated code length per candidate, adjacency
 * matrix, shortest path (reference src/SLAPredictor.c:416-468, 1521-1581, 1615-1692).  d_groups: the groups of
 * sla_hip_launch_search_exact, num_channels consecutive entries per super-frame; d_lpc_out: their (or
 * sla_hip_launch_lpc's) output.  Per super-frame: d_status 0 = d_num_parts block lengths in d_parts
 * (SLA_HIP_PLAN_NODES entries per super-frame) are exactly what the reference's host arithmetic decides;
 * 1 = a comparison came closer than the device logarithm can be trusted (or the input was flagged / not
 * finite): the caller must redo this super-frame on the host from the candidates' doubles; 2 = the super-frame's
 * candidates were certified tile sums (parcor[0] = width > 0) and some comparison came closer than the widths allow:
 * every candidate of the super-frame has been flagged (NaN in r[0]) -- the caller reruns them as serial chains
 * (sla_hip_launch_lpc_rerun) and decides on the host. */
#define SLA_HIP_PLAN_NODES 17u
int sla_hip_launch_plan(const sla_hip_lpc_group* d_groups, uint32_t num_superframes, uint32_t num_channels,
                        uint32_t order, uint32_t bits_per_sample, const sla_hip_lpc_cand* d_cands,
                        double* d_lpc_out, uint32_t* d_parts, uint32_t* d_num_parts, uint32_t* d_status,
                        sla_hip_stream_t stream);

/* Block table of one run of super-frames from sla_hip_launch_plan's partitions, written on the device, so that the
 * block-stage kernels can follow the partition search without the host building and uploading their descriptors
 * (the host part of src/SLAEncoder.c:846-869: the walk over the super-frames that numbers the blocks).
 * d_superframes: the run, in file order.  A live super-frame names its row of d_parts / d_num_parts / d_status; one
 * that is a single SILENT block (live == SLA_HIP_NOT_LIVE) takes one block index and produces no group.
 * sla_hip_launch_expand treats no block of a live super-frame as silent: for input whose non-zero mask has no all-zero
 * word (sla_hip_launch_prepass: d_or_mask[1] == 0), where a block of SLA's minimum length cannot be.
 * sla_hip_launch_expand_masked (round 4) also takes the prepass mask (d_nonzero_mask, bit s of word s / 64 = some
 * channel's sample s is not zero; NULL = the plain call): a block of a live super-frame whose samples are all zero is
 * a SILENT block (src/SLAEncoder.c:392-408) -- it takes a block index and produces no group, exactly as the host's walk
 * numbers it -- so that files WITH silence get device-written tables too.
 * d_run: 4 words, zeroed by the caller before the first run of a file; d_prefix: 2 * num_superframes words of scratch (the
 * numbering is done by one workgroup, k_expand_scan, the descriptors are written by the whole device, k_expand_write).
 * Per (block, channel), numbered on from d_run[0] blocks / d_run[1] groups: d_groups[g] (windowed form: win_off looked
 * up by block length in the d_win_len / d_win_off list, cand_first = g, slot_first = block * num_channels + channel),
 * d_cands[g] = {0, length}, d_acf_jobs[g].  counts (which may be page-locked host memory: the caller can poll
 * counts[3]): [0] blocks and [1] groups of this run, [2] 1 = tables valid, 0 = some super-frame was not certified
 * (d_status != 0), a length has no window, the tables are full or an earlier run failed (d_run[2] != 0) -- nothing
 * may be launched from them, and every later run on the same d_run fails too; [3] = sequence, written last. */
#define SLA_HIP_NOT_LIVE 0xFFFFFFFFu
typedef struct sla_hip_superframe {
  uint32_t start;          /* first sample */
  uint32_t window;         /* samples (a SILENT one: the length of the zero run) */
  uint32_t live;           /* row of the plan's arrays, or SLA_HIP_NOT_LIVE */
  uint32_t pad_;
} sla_hip_superframe;
int sla_hip_launch_expand(const sla_hip_superframe* d_superframes, uint32_t num_superframes,
                          const uint32_t* d_parts, const uint32_t* d_num_parts, const uint32_t* d_status,
                          uint32_t num_channels, uint32_t int_shift,
                          const uint32_t* d_win_len, const uint32_t* d_win_off, uint32_t num_windows,
                          uint32_t* d_run, uint32_t* d_prefix, sla_hip_lpc_group* d_groups, sla_hip_lpc_cand* d_cands,
                          sla_hip_acf_job* d_acf_jobs, uint32_t group_capacity,
                          uint32_t* counts, uint32_t sequence, sla_hip_stream_t stream);
int sla_hip_launch_expand_masked(const sla_hip_superframe* d_superframes, uint32_t num_superframes,
                          const uint32_t* d_parts, const uint32_t* d_num_parts, const uint32_t* d_status,
                          uint32_t num_channels, uint32_t int_shift,
                          const uint32_t* d_win_len, const uint32_t* d_win_off, uint32_t num_windows,
                          uint32_t* d_run, uint32_t* d_prefix, sla_hip_lpc_group* d_groups, sla_hip_lpc_cand* d_cands,
                          sla_hip_acf_job* d_acf_jobs, uint32_t group_capacity,
                          uint32_t* counts, uint32_t sequence, const uint64_t* d_nonzero_mask, sla_hip_stream_t stream);

/* Building blocks of the per-call predictor API (include/SLAPredictor.h), also usable on their own:
 *   sla_hip_launch_lpc_f64      sla_hip_launch_lpc in its search form on samples that already are doubles
 *                               (group.pcm_off indexes d_samples; no conversion, window or pre-emphasis)
 *   sla_hip_launch_lattice_raw  sla_hip_launch_lattice on samples that already are the lattice input (no shift,
 *                               no mid/side, no pre-emphasis)
 *   sla_hip_launch_tail_stages  sla_hip_launch_tail with the LMS stage optional (job.pitch = 0 switches the
 *                               long-term stage off as always); d_fold_sum still receives the zig-zag sums
 *   sla_hip_launch_emphasis_*   pre-emphasis as a pass of its own, out[n] = in[n] - ((in[n-1] * (2^s - 1)) >> s) */
int sla_hip_launch_lpc_f64(const double* d_samples, uint32_t order,
                           const sla_hip_lpc_group* d_groups, uint32_t num_groups, uint32_t max_window,
                           uint32_t max_cands_per_group, const sla_hip_lpc_cand* d_cands,
                           double* d_out, sla_hip_stream_t stream);
int sla_hip_launch_lattice_raw(const int32_t* d_samples, uint64_t plane_stride, uint32_t order,
                               const sla_hip_lattice_chunk* d_chunks, uint32_t num_chunks,
                               const int32_t* d_kint, int32_t* d_residual, sla_hip_stream_t stream);
int sla_hip_launch_tail_stages(const int32_t* d_res_in, int32_t* d_res_out, uint64_t plane_stride,
                               const sla_hip_tail_job* d_jobs, uint32_t num_jobs, uint32_t longterm_order,
                               uint32_t lms_order, uint32_t skip_lms, uint64_t* d_fold_sum, sla_hip_stream_t stream);
int sla_hip_launch_emphasis_i32(const int32_t* d_in, int32_t* d_out, uint32_t num_samples, int32_t previous,
                                uint32_t coef_shift, sla_hip_stream_t stream);
int sla_hip_launch_emphasis_f64(const double* d_in, double* d_out, uint32_t num_samples, uint32_t coef_shift,
                                sla_hip_stream_t stream);

/* Integer pre-emphasis + PARCOR lattice; one wave per chunk. */
int sla_hip_launch_lattice(const int32_t* d_pcm, uint64_t plane_stride, uint32_t mid_side, uint32_t order,
                           const sla_hip_lattice_chunk* d_chunks, uint32_t num_chunks,
                           const int32_t* d_kint, int32_t* d_residual, sla_hip_stream_t stream);

/* The same lattice, the waves derived on the device from block descriptors (one sla_hip_lpc_group per (block, channel):
 * pcm_off, num_samples, channel, int_shift, slot_first -> coefficients at d_kint[slot_first * (order + 1)]); max_window =
 * the longest block.  What the whole-file driver uses: no per-chunk descriptors to build and upload. */
int sla_hip_launch_lattice_groups(const int32_t* d_pcm, uint64_t plane_stride, uint32_t mid_side, uint32_t order,
                                  const sla_hip_lpc_group* d_groups, uint32_t num_groups, uint32_t max_window,
                                  const int32_t* d_kint, int32_t* d_residual, sla_hip_stream_t stream);
int sla_hip_launch_lattice_groups_x(const int32_t* d_pcm, uint64_t plane_stride, uint32_t mid_side, uint32_t order,
                                  const sla_hip_lpc_group* d_groups, uint32_t num_groups, uint32_t max_window,
                                  const int32_t* d_kint, int32_t* d_residual, sla_hip_stream_t stream, const sla_hip_launch_extra* extra);

/* Autocorrelation of each job's residual, computed exactly as the reference's real-FFT route does
 * (zero-padded to fft_size, forward, |.|^2, inverse).  head == SLA_HIP_ACF_RECORD: per job a 12-double
 * record {code (0 silent, 1 ok, 2 no candidate), chosen pitch lag, acf[0..4], acf[chosen-2..chosen+2]}
 * after the reference's peak scan (src/SLAPredictor.c:866-924); any other head: the first `head` lags.  d_twiddles holds the
 * SLA_HIP_TWIDDLE_DOUBLES(fft_size) doubles produced by the host with the reference's recurrence (layout: sla_kernels.hip).
 * fft_size*8 bytes must fit SLA_HIP_LDS_BUDGET, otherwise d_scratch (scratch_slots x fft_size doubles
 * of device memory) is used as the work area. */
#define SLA_HIP_ACF_RECORD 12u
#define SLA_HIP_TWIDDLE_DOUBLES(fft_size) (6u * (size_t)(fft_size))
int sla_hip_launch_ltm_acf(const int32_t* d_residual, uint64_t plane_stride,
                           const sla_hip_acf_job* d_jobs, uint32_t num_jobs, uint32_t fft_size,
                           const double* d_twiddles, double* d_scratch, uint32_t scratch_slots,
                           double* d_acf_head, uint32_t head, sla_hip_stream_t stream);
int sla_hip_launch_ltm_acf_x(const int32_t* d_residual, uint64_t plane_stride,
                           const sla_hip_acf_job* d_jobs, uint32_t num_jobs, uint32_t fft_size,
                           const double* d_twiddles, double* d_scratch, uint32_t scratch_slots,
                           double* d_acf_head, uint32_t head, sla_hip_stream_t stream, const sla_hip_launch_extra* extra);

/* Long-term pitch and taps on the device: one lane per job solves the Wiener system around the lag k_ltm_acf chose
 * (d_acf_records: num_jobs compact records, head SLA_HIP_ACF_RECORD) and writes the job k_tail reads -- position and
 * channel from d_groups[job], pitch (0: stage bypassed) and the quantised taps (src/SLAPredictor.c:855-863, 913-979;
 * src/SLAUtility.c:487-674; src/SLAEncoder.c:629-640).  The reference's long double refinement residual is computed
 * in integer arithmetic with a 64-bit significand, so the taps are the reference's bit for bit.  longterm_order: 1, 3, 5. */
int sla_hip_launch_ltm_solve(const double* d_acf_records, const sla_hip_lpc_group* d_groups, uint32_t num_jobs,
                             uint32_t longterm_order, sla_hip_tail_job* d_jobs, sla_hip_stream_t stream);

/* Long-term filter + sign-log LMS + folded-residual sum, one lane per job.
 * d_res_in/d_res_out are channel planes with the same stride as the PCM. */
int sla_hip_launch_tail(const int32_t* d_res_in, int32_t* d_res_out, uint64_t plane_stride,
                        const sla_hip_tail_job* d_jobs, uint32_t num_jobs, uint32_t longterm_order,
                        uint32_t lms_order, uint64_t* d_fold_sum, sla_hip_stream_t stream);
int sla_hip_launch_tail_x(const int32_t* d_res_in, int32_t* d_res_out, uint64_t plane_stride,
                        const sla_hip_tail_job* d_jobs, uint32_t num_jobs, uint32_t longterm_order,
                        uint32_t lms_order, uint64_t* d_fold_sum, sla_hip_stream_t stream, const sla_hip_launch_extra* extra);

/* Rice code lengths: per job the log2 of both adaptive moduli for every sample (d_kk, same plane layout
 * as the residual, uint16: k0 | k1 << 8) and the channel's total body bits (d_chan_bits[job]).
 * replaces the parameter walk of SLACoder_PutDataArray (src/SLACoder.c:224-270, 429-467). */
int sla_hip_launch_rice_len(const int32_t* d_residual, uint64_t plane_stride, const sla_hip_rice_job* d_jobs,
                            uint32_t num_jobs, uint16_t* d_kk, uint64_t* d_chan_bits, sla_hip_stream_t stream);

/* Block assembly into a zero-initialised .sla image (32-bit words, 4-byte aligned at file offset 0):
 * pre-packed header bytes, channel-interleaved Rice/Golomb/gamma or RAW bodies, then CRC16 + size patch.
 * replaces src/SLAEncoder.c:740-798 and src/SLACoder.c:45-82,120-138,429-467. */
int sla_hip_launch_rice_write(const int32_t* d_residual, const int32_t* d_pcm, uint64_t plane_stride,
                              const uint16_t* d_kk, const sla_hip_pack_block* d_blocks, uint32_t num_blocks,
                              const uint8_t* d_headers, uint32_t num_channels, uint32_t raw_shift,
                              uint32_t mid_side, uint32_t* d_image, sla_hip_stream_t stream);

/* int16 -> left-justified int32 (<< 16); d_in must be 8-byte aligned.  Used by the host-PCM path of
 * SLAEncoder_EncodeWhole to halve the PCIe bytes of <= 16-bit input. */
int sla_hip_launch_unpack16(const int16_t* d_in, int32_t* d_out, uint64_t count, sla_hip_stream_t stream);
/* the same for <= 24 significant bits carried as three bytes per sample (little-endian, the int32's upper three bytes) */
int sla_hip_launch_unpack24(const uint8_t* d_in, int32_t* d_out, uint64_t count, sla_hip_stream_t stream);

/* ---- decode side (kernels in sla_decode.hip, driver: SLADecoder.h) -----------------
 * The stream image is the .sla file's bytes in device memory (hipMalloc alignment, viewed as 32-bit words).
 * The host walks the block chain (sync code, size field, sample count: 10 bytes per block) into a table of
 * sla_hip_dec_block; everything behind those fields is parsed on the device.  All stages work in place on
 * one set of channel planes [C][plane_stride] int32:
 *   dec_bits    : block header fields + entropy decode (src/SLADecoder.c:355-412, 440-481; src/SLACoder.c:84-163,
 *                 272-318, 406-427, 469-506) -> right-justified residual / raw samples / zeros; per block
 *                 sla_hip_dec_info, per (block, channel) sla_hip_dec_chan and the PARCOR coefficients
 *                 d_kint[(block * C + ch) * (order + 1) + m]; with want_crc also the CRC16 of every block
 *   dec_lms     : SLALMSFilter_SynthesizeInt32              src/SLAPredictor.c:1334-1463
 *   dec_ltm     : SLALongTermSynthesizer_SynthesizeInt32    src/SLAPredictor.c:1034-1119
 *   dec_lattice : SLALPCSynthesizer_SynthesizeByParcorCoefInt32 src/SLAPredictor.c:610-740
 *                 + SLAEmphasisFilter_DeEmphasisInt32       src/SLAPredictor.c:1768-1791 (deemphasis != 0)
 *   dec_finish  : SLAUtility_MStoLRInt32 src/SLAUtility.c:415-433 + left-justification src/SLADecoder.c:540-547
 * The synthesis kernels skip blocks whose decoded type is not "compressed". */
#define SLA_HIP_DEC_HEADER_ONLY 1u       /* sla_hip_dec_block.flags: parse the header (and CRC), decode no samples */
typedef struct sla_hip_dec_block {
  uint64_t byte_off;        /* offset of the block's sync code in the image */
  uint32_t byte_len;        /* size field + 6, clipped to the end of the stream (CRC16 covers [8, byte_len)) */
  uint32_t smp_off;         /* first sample of the block in the planes */
  uint32_t num_samples;     /* samples per channel, from the block header */
  uint32_t flags;
} sla_hip_dec_block;
typedef struct sla_hip_dec_info {
  uint32_t type;            /* 0 compressed, 1 silent, 2 raw, 3 not a block type */
  uint32_t used_bytes;      /* bytes from the sync code to the byte-aligned end of what the reader consumed */
  uint32_t crc;             /* CRC16-IBM of [8, byte_len) when requested */
  uint32_t overrun;         /* 1: the reader ran off the end of the stream */
} sla_hip_dec_info;
typedef struct sla_hip_dec_chan {
  uint32_t pitch;           /* 0: no long-term stage */
  int32_t  ltm_coef[5];     /* Q31 */
  uint32_t rice_init;
  uint32_t reserved;
} sla_hip_dec_chan;
int sla_hip_launch_dec_bits(const uint32_t* d_image, uint64_t image_bytes,
                            const sla_hip_dec_block* d_blocks, uint32_t num_blocks,
                            uint32_t num_channels, uint32_t bits_per_sample, uint32_t offset_lshift,
                            uint32_t mid_side, uint32_t parcor_order, uint32_t longterm_order,
                            uint32_t want_crc, int32_t* d_planes, uint64_t plane_stride,
                            sla_hip_dec_info* d_info, sla_hip_dec_chan* d_chan, int32_t* d_kint,
                            sla_hip_stream_t stream);
int sla_hip_launch_dec_lms(int32_t* d_planes, uint64_t plane_stride, const sla_hip_dec_block* d_blocks,
                           const sla_hip_dec_info* d_info, uint32_t num_blocks, uint32_t num_channels,
                           uint32_t lms_order, sla_hip_stream_t stream);
int sla_hip_launch_dec_ltm(int32_t* d_planes, uint64_t plane_stride, const sla_hip_dec_block* d_blocks,
                           const sla_hip_dec_info* d_info, const sla_hip_dec_chan* d_chan,
                           uint32_t num_blocks, uint32_t num_channels, uint32_t longterm_order,
                           uint32_t max_block_samples, sla_hip_stream_t stream);
int sla_hip_launch_dec_lattice(int32_t* d_planes, uint64_t plane_stride, const sla_hip_dec_block* d_blocks,
                               const sla_hip_dec_info* d_info, uint32_t num_blocks, uint32_t num_channels,
                               const int32_t* d_kint, uint32_t parcor_order, uint32_t deemphasis, sla_hip_stream_t stream);
/* de-emphasis alone (per-call API): in place, y[-1] = previous */
int sla_hip_launch_dec_deemphasis(int32_t* d_data, uint32_t num_samples, int32_t previous, uint32_t coef_shift,
                                  sla_hip_stream_t stream);
int sla_hip_launch_dec_finish(int32_t* d_planes, uint64_t plane_stride, uint32_t num_channels,
                              uint32_t num_samples, uint32_t mid_side, uint32_t shift, sla_hip_stream_t stream);

/* ---- (2) whole-file driver ---------------------------------------------- */

/* Per-block results of the last analyze call, copied into caller arrays
 * (layout identical to the oracle's trace so tests can diff them). */
typedef struct sla_hip_trace {
  uint32_t  max_blocks;      /* in  */
  uint32_t  order_stride;    /* in: parcor_order + 1 */
  uint32_t  ltm_stride;      /* in: longterm_order   */
  uint32_t  sample_stride;   /* in: per-channel stride of res_* (>= num_samples) */
  uint32_t  num_blocks;      /* out */
  uint32_t  offset_lshift;   /* out */
  uint32_t* blk_start; uint32_t* blk_nsmpl; uint32_t* blk_type; uint32_t* blk_bytes;
  double*   parcor; int32_t* code; int32_t* kint;
  uint32_t* rshift; uint32_t* pitch; int32_t* ltm_coef; uint32_t* rice_init;
  int32_t*  res_lattice; int32_t* res_final;    /* may be NULL: not copied back */
  uint32_t* parcor_exact;    /* may be NULL; per (block, channel): 1 = parcor[] are the reference's doubles bit for bit (exact
                                chain kernel), 0 = certified: code[], kint[] and the RAW decision are the reference's, the
                                doubles agree to the certificate's bound (option "block_cert") */
} sla_hip_trace;

/* Hot path on PCM resident in device memory: planar int32 [C][plane_stride],
 * left-justified.  On return the encoder holds, on the device, the final
 * residual planes and, on the host, the block table and per-block parameters.
 * `timing_ms` (may be NULL) receives 12 floats [ms]: HIP-event durations of k_prepass, k_lpc (search),
 * k_lpc (blocks), k_lattice, k_tail; host wall time of planning and of the long-term solve; total
 * wall time; HIP-event duration of k_ltm_acf; number of pipeline chunks; search groups that had to rerun
 * as serial chains; 1 if the search ran on tile sums (sla_hip_launch_search_exact), else 0. */
int sla_hip_analyze_device(struct SLAEncoder* encoder, const int32_t* d_pcm, uint64_t plane_stride,
                           uint32_t num_samples, sla_hip_stream_t stream, float* timing_ms);

/* Bit-serial pack of the analysed file into `data` (host): D2H of the final
 * residual, block headers, Rice body, CRC16 (reference src/SLAEncoder.c:682-798,
 * src/SLACoder.c:429-467).  Needs a preceding sla_hip_analyze_device. */
int sla_hip_pack(struct SLAEncoder* encoder, uint8_t* data, uint32_t data_size, uint32_t* output_size);

/* Same bytes as sla_hip_pack, but the Rice coding, block assembly and CRC16 run on the device and one
 * D2H copy brings the finished image back.  Used by SLAEncoder_EncodeWhole. */
int sla_hip_pack_device(struct SLAEncoder* encoder, uint8_t* data, uint32_t data_size, uint32_t* output_size);

/* Many files in one call (BASELINE C4: a batch of short clips).  Each file is encoded exactly as
 * SLAEncoder_EncodeWhole would encode it on its own -- same bytes, own header, own offset_lshift -- but the blocks
 * of all files travel through the kernels together (they are as independent as the blocks of one file), so a
 * 10-second clip no longer costs the fixed latency of the whole pipeline.  All files share the encoder's wave format
 * and parameters.  Returns 0 when the batch ran; each item carries its own SLAApiResult (e.g. a buffer too small). */
typedef struct sla_hip_batch_item {
  const int32_t* const* input;     /* in : [num_channels] planes, left-justified in 32 bits */
  uint32_t num_samples;            /* in  */
  uint32_t data_size;              /* in : capacity of `data` */
  uint8_t* data;                   /* in : receives the .sla bytes */
  uint32_t output_size;            /* out */
  int32_t  result;                 /* out: SLAApiResult of this file */
} sla_hip_batch_item;
int sla_hip_encode_batch(struct SLAEncoder* encoder, sla_hip_batch_item* items, uint32_t num_items);
/* The analysis of such a batch when the planes already live in device memory ([C][plane_stride] int32, files at
 * file_start[i] -- multiples of SLA_HIP_PREPASS_TILE, ascending, gaps zero -- of file_samples[i] samples, all inside
 * [0, span)): the batch counterpart of sla_hip_analyze_device.  file_lshift (may be NULL) receives every file's
 * offset_lshift; files that share a value share a pipeline pass.  After a single pass sla_hip_get_trace describes the
 * blocks of all files (positions relative to the planes). */
int sla_hip_analyze_batch_device(struct SLAEncoder* encoder, const int32_t* d_pcm, uint64_t plane_stride, uint32_t span,
                                 const uint32_t* file_start, const uint32_t* file_samples, uint32_t num_files,
                                 uint32_t* file_lshift, float* timing_ms);

/* ---- one file, several GPUs ------------------------------------------------------------------------------
 * Blocks are independent (every filter and the coder reset per block, src/SLAEncoder.c:594-659), so the super-frames
 * of ONE file shard over the GPUs of a node, one process and one encoder handle per GPU.  Three facts of the whole
 * file have to be agreed on first, and the library leaves the exchange to the caller (RCCL from C, or
 * torch.distributed as in sla_amd/dist.py -- the library itself links no collective library):
 *   offset_lshift   comes from the OR of EVERY sample (src/SLAEncoder.c:425-455)      -> OR of 4 bytes per rank (all-gathered: RCCL has no bitwise reduction)
 *   super-frames    hop over silence runs, so where one starts depends on everything before it
 *                   (src/SLAEncoder.c:392-408, 846-869)                                -> all-gather of the 1-bit mask
 *   the header      counts blocks and keeps the largest block / bit rate (:920-926)   -> gathered with the bytes
 * Sequence on rank r of `world` (file of N samples per channel, pieces cut at multiples of 1024):
 *   1. scan pieces tile [0, N): piece r = [cut(r), cut(r+1)), cut(r) = ceil(N*r/world) floored to a multiple of 1024.
 *      Upload [cut(r), min(N, cut(r+1) + 1023 + max_num_block_samples)): bounds[r+1] of step 3 is the first
 *      super-frame start at or behind the UNFLOORED target ceil(N*(r+1)/world), i.e. less than one maximum block
 *      behind it (sla_amd/dist.py: upload_range);
 *      sla_hip_shard_scan on exactly the piece  ->  OR word, mask bits of the piece
 *   2. all-gather the OR words and OR them, all-gather the mask pieces (N/8 bytes in total)
 *   3. sla_hip_shard_bounds (pure host arithmetic, every rank computes the same table): rank r owns
 *      [bounds[r], bounds[r+1]), both super-frame starts of the whole file's hop
 *   4. sla_hip_shard_analyze on the planes of that range with the file's OR word (the hot path: exactly
 *      sla_hip_analyze_device, but offset_lshift and the sample unit are the file's), then sla_hip_pack_device:
 *      a complete .sla image of the range, 43-byte header + blocks
 *   5. gather sizes and bytes on one rank (all-gather over xGMI: compressed bytes, less than the residual planes the
 *      north star's variant gathers -- sla_amd/dist.py keeps both); that rank drops the headers of ranks > 0 and
 *      writes the file's header with sla_hip_shard_header.
 * The result is byte-identical to SLAEncoder_EncodeWhole of the whole file on one GPU. */
int sla_hip_shard_scan(struct SLAEncoder* encoder, const int32_t* d_pcm, uint64_t plane_stride, uint32_t num_samples,
                       uint32_t* or_word, uint64_t* nz_mask /* host, ceil(num_samples / 64) words */);
/* Steps 1-3 without the mask when the file has no silence (the usual case): the scan only brings home the OR word and
 * the number of all-zero 64-sample mask words of the piece; a silence run moves a super-frame start only when it is at
 * least 2048 samples long, so if NO rank counts an all-zero word the hop is plain (nz_mask = NULL below) and the
 * ranks all-gather 12 bytes each (OR word as two int32 halves + the count) instead of N/8 in total.  Otherwise fall back to sla_hip_shard_scan + the mask. */
int sla_hip_shard_scan_counts(struct SLAEncoder* encoder, const int32_t* d_pcm, uint64_t plane_stride, uint32_t num_samples,
                              uint32_t* or_word, uint32_t* zero_mask_words);
int sla_hip_shard_bounds(uint32_t num_samples, uint32_t max_num_block_samples, const uint64_t* nz_mask /* NULL: no silence */,
                         uint32_t world, uint32_t* bounds /* world + 1 entries */);
int sla_hip_shard_analyze(struct SLAEncoder* encoder, const int32_t* d_pcm, uint64_t plane_stride, uint32_t num_samples,
                          uint32_t file_or_word, float* timing_ms);
/* sla_hip_shard_analyze when the counts of step 1 showed no all-zero mask word anywhere in the file (and its OR word is
 * not 0): nothing the range's own prepass could find, so it is skipped (one kernel and one host wait per call less).
 * The caller vouches for the counts; a range WITH silence analysed through this call gives a valid but different file. */
int sla_hip_shard_analyze_no_silence(struct SLAEncoder* encoder, const int32_t* d_pcm, uint64_t plane_stride, uint32_t num_samples,
                                     uint32_t file_or_word, float* timing_ms);
int sla_hip_shard_header(const uint8_t* const* shard_headers /* world x 43 bytes */, uint32_t world, uint8_t* data, uint32_t data_size);

/* Device pointers of the last analysis (for RCCL gathers / tests). */
const int32_t* sla_hip_final_residual(const struct SLAEncoder* encoder, uint64_t* plane_stride);
const int32_t* sla_hip_lattice_residual(const struct SLAEncoder* encoder, uint64_t* plane_stride);

/* Make the analysis write its residual planes into caller-owned device memory ([C][plane_stride]
 * int32 each; plane_stride must equal the one later passed to sla_hip_analyze_device).  Used when
 * the planes feed a collective (RCCL all-gather of the residual stream).  NULL, NULL unbinds. */
int sla_hip_bind_residual_planes(struct SLAEncoder* encoder, int32_t* d_lattice, int32_t* d_final,
                                 uint64_t plane_stride);

/* Copy the last analysis into caller arrays. */
int sla_hip_get_trace(struct SLAEncoder* encoder, sla_hip_trace* trace);

/* Options of one encoder handle, by name.  A handle starts with the defaults below; nothing is read from the environment
 * (SLA_HIP_TRACE=1 alone, at SLAEncoder_Create: a host-side timeline of every analysis on stderr).
 * Layout knobs: "lpc_pack", "lpc_threads" (256 / 512), "lpc_tile" (24 / 48 / 0 = automatic), "tail_waves" (waves per tail
 * workgroup, 1..4, 0 = 4), "tail_taps" (k_tailk: taps of each history per lane, 1 / 2 / 4, 0 = by the number of jobs), "chunks"
 * (pipeline chunks, 1..8), "threads" (host pool; default min(6, CPUs of the process): a one-process-per-GPU launcher that
 * shares few cores between its ranks sets it), "rice_lanes" (Rice parameter walk of the device pack: 0 = by the number of
 * jobs, 1 = one lane per job, 2 = two-lane pipeline), "lattice_plain" (1: every lattice stage in the wrapping
 * four-instruction form).
 * Route switches -- every route gives the same bytes; tests force the slower exact ones through these: "search_exact" (0: no
 * tile-sum search), "exact_bits" (log2 of the tile-sum energy limit, 1..53), "cert_safety" (safety factor of the certificate
 * for windows over that limit, default 64; 0: no certificate, such windows are rerun as serial chains), "device_plan" (0:
 * partitions decided on the host), "block_cert" (0: every chosen block through the exact chain kernel; 1 = default: the
 * certified route of sla_hip_launch_lpc_blocks_cert), "block_cert_safety" (default 16), "cert_audit" (N > 0: every N-th
 * certified pair is re-analysed by the exact kernels and compared, a difference fails the call; default 0), "plan_margin",
 * "lpc_blocks_chains", "fuse_lattice", "device_ltm" (0: long-term pitch + taps solved on the host threads from the downloaded
 * autocorrelations, one tail launch per pipeline chunk), "single_tail" (0: one tail launch per pipeline chunk instead of one
 * for the file), "first_chunk" (1/1000 of the super-frames in pipeline chunk 0; 0: built-in shares), "alt_streams" (block
 * stages of odd and even pipeline chunks on two streams: 0 never, 1 / 2 = default: whenever the file is cut into chunks),
 * "device_expand" (1 = default: block tables of certified partitions written on the device, sla_hip_launch_expand, the
 * host's copy following under the kernels; 0: host tables first), "expand_silence" (1 = default: input with silence takes
 * the device tables too, sla_hip_launch_expand_masked; 0: host tables for such input), "table_cache" (1 = default: the search tables of a file --
 * or batch -- without silence are kept for the next one of the same layout and parameters), "prelaunch" (1 = default: short
 * files queue the certified block kernels together with the searches, sized for the most groups there can be, the kernels
 * reading the number from the device), "one_stream" (1: a one-chunk file keeps search, block stage and tail on one stream;
 * measured slower, default 0), "upload24" (1 = default: pageable input of 17..24 significant bits crosses the bus as three
 * bytes per sample; DESIGN section 7 has the A/B).
 * SLAEncoder_EncodeWhole of long files: "stream" (0: never streamed), "stream_piece" (samples per piece, all channels
 * together; default 32 Mi; a file of fewer than two pieces is not streamed), "stream_lanes" (worker lanes, 1..6, default 6; pieces are handed to whichever lane is free),
 * "batch_lanes" (sla_hip_encode_batch: a batch of at least 8 files and 16 Mi samples is dealt out in groups of consecutive files to
 * that many worker lanes, uploads taking turns, everything behind them overlapping; 1..6, default 4; 1 = the batch in one piece).
 * After a streamed call the handle holds no analysis tables: sla_hip_get_trace / sla_hip_pack / sla_hip_final_residual answer
 * SLA_APIRESULT_PARAMETER_NOT_SET (NULL).
 * Returns SLA_APIRESULT_INVALID_ARGUMENT for an unknown name or a value out of range. */
int sla_hip_encoder_set_option(struct SLAEncoder* encoder, const char* name, double value);

/* Name of the device the library bound to ("" when none). */
const char* sla_hip_device_name(void);

/* The 12 floats of the last analysis (sla_hip_analyze_device or SLAEncoder_EncodeWhole/Block). */
int sla_hip_last_timing(const struct SLAEncoder* encoder, float* timing_ms);

/* 6 counters of the last analysis: search groups rerun as serial chains; super-frames whose partition the host
 * had to decide (device plan not certified); 1 if the search ran on tile sums; 1 if the device plan is enabled;
 * k_tail launches (1: one for the file, else one per pipeline chunk); 1 if pitch + taps were solved on the device. */
int sla_hip_last_counters(const struct SLAEncoder* encoder, uint32_t* counters);

/* 2 counters of the last analysis: 1 if the block stage took the certified route (sla_hip_launch_lpc_blocks_cert);
 * (block, channel) pairs its certificate handed to the exact chain kernels. */
int sla_hip_last_block_cert(const struct SLAEncoder* encoder, uint32_t* counters);
/* 2 counters of the last analysis under option "cert_audit": certified pairs the exact kernels re-analysed and found equal;
 * pairs they found DIFFERENT (the analysis that saw one returned SLA_APIRESULT_NG). */
int sla_hip_last_cert_audit(const struct SLAEncoder* encoder, uint32_t* counters);

/* 4 counters: pipeline chunks of the last analysis whose block stage was launched from device-written tables
 * (sla_hip_launch_expand; option "device_expand"), its pipeline chunks in all; since the handle was created: the analyses
 * that found their search tables kept from the file before (same length and parameters, no silence; option
 * "table_cache"), and those among them whose searches -- launched behind the prepass on the guess that the file is like
 * the last one (short files only) -- had to go out again because the prepass said otherwise. */
int sla_hip_last_expand(const struct SLAEncoder* encoder, uint32_t* counters);

/* 4 floats [ms]: execution time of k_lpc_blocks, k_lattice, k_ltm_acf, k_tail in the last analysis, summed over
 * its launches and measured ON the device (first wave in to last wave out, constant 100 MHz clock) -- what
 * rocprofv3 --kernel-trace reports, without the time a launch spends queued behind kernels of other streams
 * that a pair of stream events includes. */
int sla_hip_last_kernel_ms(const struct SLAEncoder* encoder, float* kernel_ms);

#ifdef __cplusplus
}
#endif

#endif /* SLA_HIP_H_INCLUDED */
