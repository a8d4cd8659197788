// sla_kernels.hip -- gfx950 (MI355X / CDNA4) kernels of the SLA encode hot path.
//
// Written for 64-lane wavefronts, 160 KiB LDS per CU, no MFMA (there is no dense
// contraction anywhere on this path: FP64 VALU chains + int32 VALU).  Built with
// -ffp-contract=off: every FP64 product and sum below is rounded separately, in the
// order written, because the PARCOR codes that reach the bit stream come from
// Round(k * 2^(q-1)) of these doubles (reference src/SLAEncoder.c:578-582) and the
// reference x86-64 build has no FMA.
//
// Kernel (file under kernels/)           replaces (reference file:line)
//   k_prepass, k_batch_scan (prepass)     src/SLAEncoder.c:425-455 (OR of all words), :392-408 / :520-528 (silence)
//   k_acf_tiles[_lds], k_search_finish,   src/SLAPredictor.c:1615-1649 (partition search) over :331-388 (autocorrelation): tile sums
//   k_search_cert (search)                where the order of summation provably cannot matter, certified where it can; :253-328
//   k_lpc (lpc_chains)                    src/SLAPredictor.c:331-388 in the reference's serial order + :253-328: the rerun of windows
//                                         the certificate could not decide, and the per-call predictor API
//   k_plan (search)                       src/SLAPredictor.c:416-468 (code length), :1521-1581 (Dijkstra), :1652-1692 (partition),
//                                         certified against libm's last bits
//   k_expand_scan / _write (expand)       src/SLAEncoder.c:846-869 (the block table of the partitions)
//   k_acf_blocks (search),                src/SLAEncoder.c:505-515,540-543 (staging), src/SLAUtility.c:370-412 (mid/side),
//   k_blocks_finish, k_lpc_blocks         src/SLAPredictor.c:331-388 (autocorrelation: any order + certificate, or the exact term
//   (blocks)                              tiles), :253-328 (Levinson-Durbin), src/SLAEncoder.c:567-589 (quantiser) of the chosen blocks
//   k_lattice, k_lattice_groups           src/SLAPredictor.c:1741-1765 (pre-emphasis), :557-607 (PARCOR lattice; stage forms in
//   (lattice, lattice_wave)               lattice_wave.inc)
//   k_ltm_acf, k_ltm_acf2, k_ltm_solve    src/SLAPredictor.c:827-979 (long-term analysis: FFT autocorrelation, pitch, taps),
//   (longterm)                            src/SLAUtility.c:219-319 (four1 / realft), :449-674 (LU + refinement)
//   k_tailk (tail)                        src/SLAPredictor.c:1031-1119 (long-term filter), :1202-1331 (sign-log LMS),
//                                         src/SLACoder.c:361-385 (mean of folded residual)
//   k_rice_k/_k2/_bits/_write,            src/SLACoder.c:45-83,120-139,165-271,388-467 (Golomb, gamma, recursive Rice, PutDataArray),
//   k_block_crc, k_unpack16/24 (pack)     src/SLAEncoder.c:682-798 (block assembly), src/SLAUtility.c:321-339 (CRC16)
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <float.h>
#include <stdlib.h>
#include <pthread.h>
#include <stdio.h>
#include <string.h>

#include "sla_hip.h"
#include "sla_crc_dev.h"

#define SLA_WAVE 64

// ---------------------------------------------------------------------------------------------
// shared helpers
// ---------------------------------------------------------------------------------------------

// right-justified integer sample of output channel `ch` (mid/side when ms != 0)
__device__ __forceinline__ int32_t load_int(const int32_t* __restrict__ pcm, uint64_t stride, uint32_t ms,
                                            uint32_t ch, uint64_t idx, uint32_t shift)
{
  if (!ms) { return pcm[(uint64_t)ch * stride + idx] >> shift; }
  int32_t l = pcm[idx] >> shift, r = pcm[stride + idx] >> shift;
  // mid = (L+R)>>1 (arithmetic, wrapping sum), side = L-R      src/SLAUtility.c:403-411
  return (ch == 0) ? ((int32_t)((uint32_t)l + (uint32_t)r) >> 1) : (int32_t)((uint32_t)l - (uint32_t)r);
}

// analysis sample: (double)in * 2^-31, mid = (l+r)/2, side = l-r   src/SLAEncoder.c:507, src/SLAUtility.c:382-387
__device__ __forceinline__ double load_f64(const int32_t* __restrict__ pcm, uint64_t stride, uint32_t ms,
                                           uint32_t ch, uint64_t idx)
{
  const double scale = 4.656612873077392578125e-10;   // 2^-31, exact
  if (!ms) { return (double)pcm[(uint64_t)ch * stride + idx] * scale; }
  double l = (double)pcm[idx] * scale, r = (double)pcm[stride + idx] * scale;
  return (ch == 0) ? ((l + r) / 2) : (l - r);
}

// execution span of a launch, measured on the device: span[0] = max(~start), span[1] = max(end) in ticks of the
// constant 100 MHz clock (zero-initialised by the host; NULL = not wanted).  Unlike a pair of stream events this
// does not include the time a launch waits for resources behind kernels of other streams.  Only the first
// workgroup reports a start and only the first and the last one an end: workgroups are dispatched in order, the
// last one out is (within one workgroup's run time) the last one in -- and one atomic per WAVE on one address
// would serialise a 60k-wave launch.
__device__ __forceinline__ void span_begin(unsigned long long* span)
{
  if (span != nullptr && blockIdx.x == 0 && threadIdx.x == 0) { atomicMax(&span[0], ~(unsigned long long)wall_clock64()); }
}
__device__ __forceinline__ void span_end(unsigned long long* span)
{
  if (span != nullptr && (blockIdx.x == 0 || blockIdx.x == gridDim.x - 1) && (threadIdx.x & 63) == 0) {
    atomicMax(&span[1], (unsigned long long)wall_clock64());
  }
}

__device__ __forceinline__ uint32_t umax_wave(uint32_t v)
{
  for (int off = 32; off > 0; off >>= 1) { uint32_t o = __shfl_xor(v, off); v = (o > v) ? o : v; }
  return v;
}

// One translation unit, cut by stage (the file had grown to 4 700 lines); the order matters: later
// parts use device functions of earlier ones.
#include "kernels/prepass.inc"
#include "kernels/lpc_chains.inc"
#include "kernels/lattice_wave.inc"
#include "kernels/blocks.inc"
#include "kernels/search.inc"
#include "kernels/expand.inc"
#include "kernels/lattice.inc"
#include "kernels/tail.inc"
#include "kernels/longterm.inc"
#include "kernels/launchers.inc"
#include "kernels/pack.inc"
