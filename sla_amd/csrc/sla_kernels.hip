// sla_kernels.hip -- gfx950 (MI355X / CDNA4) kernels of the SLA encode hot path.
//
// Written for 64-lane wavefronts, 160 KiB LDS per CU, no MFMA (there is no dense
// contraction anywhere on this path: FP64 VALU chains + int32 VALU).  Built with
// -ffp-contract=off: every FP64 product and sum below is rounded separately, in the
// order written, because the PARCOR codes that reach the bit stream come from
// Round(k * 2^(q-1)) of these doubles (reference src/SLAEncoder.c:578-582) and the
// reference x86-64 build has no FMA.
//
// Kernel                      replaces (reference file:line)
//   k_prepass                 src/SLAEncoder.c:425-455 (OR of all words), :392-408 / :520-528 (silence)
//   k_acf_tiles,              src/SLAPredictor.c:1615-1649 (partition search) over :331-388 (autocorrelation): exact
//   k_search_finish           tile sums where the order of summation provably cannot matter, :253-328 (Levinson-Durbin)
//   k_lpc                     src/SLAPredictor.c:331-388 in the reference's serial order + :253-328: the rerun of windows
//                             the tile-sum search flagged, and the original path of every stage (debug switches)
//   k_plan                    src/SLAPredictor.c:416-468 (code length), :1521-1581 (Dijkstra), :1652-1692 (partition),
//                             certified against libm's last bits
//   k_lpc_blocks              src/SLAEncoder.c:505-515,540-543 (staging), src/SLAUtility.c:370-412 (mid/side),
//                             src/SLAPredictor.c:331-388 (autocorrelation, term tiles), :253-328 (Levinson-Durbin),
//                             src/SLAEncoder.c:567-589 (quantiser) for the chosen blocks
//   k_lattice                 src/SLAPredictor.c:1741-1765 (pre-emphasis), :557-607 (PARCOR lattice)
//   k_ltm_acf                 src/SLAPredictor.c:827-924 (long-term analysis: FFT autocorrelation, pitch candidates),
//                             src/SLAUtility.c:219-319 (four1 / realft)
//   k_tail                    src/SLAPredictor.c:1031-1119 (long-term filter), :1202-1331 (sign-log LMS),
//                             src/SLACoder.c:361-385 (mean of folded residual)
//   k_rice_len, k_rice_write, src/SLACoder.c:45-83,120-139,165-271,388-467 (Golomb, gamma, recursive Rice, PutDataArray),
//   k_block_crc               src/SLAEncoder.c:682-798 (block assembly), src/SLAUtility.c:321-339 (CRC16)
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

// ---------------------------------------------------------------------------------------------
// k_prepass: one 64-bit "non-zero" word per 64 samples, OR of every raw word, count of all-zero words.
// A wave owns 16 consecutive mask words = 1024 samples and reads them fully coalesced, four words'
// worth of samples in flight at a time (the kernel is a pure HBM stream: 4 B per sample and channel).
// The count lets the host skip the copy of the mask when the file has no silence to find: a silent run
// long enough to become a block (>= 2048 samples) contains all-zero words.
// ---------------------------------------------------------------------------------------------
#define PREPASS_WORDS 16    // mask words per wave: enough waves in flight to cover the HBM latency
template <int NCH>      // 1, 2 or 0 = any
__global__ __launch_bounds__(256)
void k_prepass(const int32_t* __restrict__ pcm, uint64_t stride, uint32_t nch_rt, uint32_t n,
               uint32_t shift, uint32_t ms, uint32_t* __restrict__ or_mask, uint64_t* __restrict__ nz_mask,
               uint32_t* __restrict__ tile_or)
{
  const uint32_t nch = (NCH != 0) ? (uint32_t)NCH : nch_rt;
  const uint32_t lane = threadIdx.x & 63;
  const uint64_t wave = (uint64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  const uint64_t nwords = ((uint64_t)n + 63) / 64;
  uint32_t acc = 0, zero_words = 0;
  // (one or two channels: 16 mask words = 4 KB requested before the first one is looked at -- the kernel is bound by how
  // many bytes the chip has in flight: C2 80 -> 68 us with 8 words)
  constexpr uint32_t WSTEP = (NCH == 1) ? 16u : (NCH == 2) ? 8u : 4u;
  for (uint32_t w0 = 0; w0 < PREPASS_WORDS; w0 += WSTEP) {
    const uint64_t word0 = wave * PREPASS_WORDS + w0;
    if (word0 >= nwords) { break; }
    if (NCH == 1 || NCH == 2) {
      int32_t a[WSTEP], b[WSTEP];
#pragma unroll
      for (int u = 0; u < (int)WSTEP; u++) {
        const uint64_t idx = (word0 + u) * 64 + lane;
        const bool in = (idx < n);
        a[u] = in ? pcm[idx] : 0;
        b[u] = (NCH == 2 && in) ? pcm[stride + idx] : 0;
      }
#pragma unroll
      for (int u = 0; u < (int)WSTEP; u++) {
        acc |= (uint32_t)a[u] | (uint32_t)b[u];
        bool nz;
        if (NCH == 2 && ms) {
          const int32_t l = a[u] >> shift, r = b[u] >> shift;
          nz = (((int32_t)((uint32_t)l + (uint32_t)r) >> 1) != 0) || ((uint32_t)l != (uint32_t)r);
        } else {
          nz = ((a[u] >> shift) != 0) || ((b[u] >> shift) != 0);
        }
        const uint64_t bits = __ballot(nz);
        if (word0 + u < nwords) {
          if (lane == 0) { nz_mask[word0 + u] = bits; }
          zero_words += (bits == 0) ? 1u : 0u;
        }
      }
    } else {
      for (uint32_t u = 0; u < 4; u++) {
        const uint64_t idx = (word0 + u) * 64 + lane;
        bool nz = false;
        if (idx < n) {
          for (uint32_t c = 0; c < nch; c++) { acc |= (uint32_t)pcm[(uint64_t)c * stride + idx]; }
          for (uint32_t c = 0; c < nch; c++) { nz = nz || (load_int(pcm, stride, ms, c, idx, shift) != 0); }
        }
        const uint64_t bits = __ballot(nz);
        if (word0 + u < nwords) {
          if (lane == 0) { nz_mask[word0 + u] = bits; }
          zero_words += (bits == 0) ? 1u : 0u;
        }
      }
    }
  }
  for (int off = 32; off > 0; off >>= 1) { acc |= __shfl_xor(acc, off); }
  // thousands of waves, one word: only the few that still add a bit pay for the atomic (a stale read just costs one)
  if (lane == 0 && (acc & ~__atomic_load_n(or_mask, __ATOMIC_RELAXED)) != 0) { atomicOr(or_mask, acc); }
  if (lane == 0 && zero_words != 0) { atomicAdd(or_mask + 1, zero_words); }
  if (tile_or != nullptr && lane == 0) { tile_or[wave] = acc; }       // OR of this wave's 1024 samples (batches: offset_lshift per file)
}

// ---------------------------------------------------------------------------------------------
// k_lpc: autocorrelation chains + Levinson-Durbin (+ quantiser) per LDS-staged window
//
// LDS: x[x_region] doubles | r[cands*(order+1)];  x_region >= max(W, 2*cands*(order+2)) because the
// Levinson work vectors a[], v[] overlay the sample window once the chains are done.
// One thread owns one (candidate, lag) chain at a time: the sum over a lag is a strictly
// sequential FP64 accumulation (the reference's order decides the rounding), so the available
// parallelism is (candidates x lags x groups), not the samples of one sum.
// ---------------------------------------------------------------------------------------------
// Chains are latency-bound (LDS read -> add -> mul -> add, few waves per SIMD because the window owns
// the LDS), so both loops fetch CHAIN_U steps of operands before they run the CHAIN_U dependent
// accumulations: the sum order is untouched, only the loads move up.
#define CHAIN_U 8

__device__ __forceinline__ double chain_lag0(const double* __restrict__ xs, uint32_t n)
{
  // software pipelined: the next CHAIN_U operands are requested before the current ones are summed (the
  // LDS queue of a CU is kept busy by the gather loads of the lag chains, so a request can take long)
  double acc = 0.0;
  uint32_t i = 0;
  if (n >= CHAIN_U) {
    double v[CHAIN_U];
#pragma unroll
    for (int u = 0; u < CHAIN_U; u++) { v[u] = xs[u]; }
    for (i = CHAIN_U; i + CHAIN_U <= n; i += CHAIN_U) {
      double w[CHAIN_U];
#pragma unroll
      for (int u = 0; u < CHAIN_U; u++) { w[u] = xs[i + u]; }
#pragma unroll
      for (int u = 0; u < CHAIN_U; u++) { acc += v[u] * v[u]; }
#pragma unroll
      for (int u = 0; u < CHAIN_U; u++) { v[u] = w[u]; }
    }
#pragma unroll
    for (int u = 0; u < CHAIN_U; u++) { acc += v[u] * v[u]; }
  }
  for (; i < n; i++) { const double v = xs[i]; acc += v * v; }
  return acc;
}

// lag >= 1, lag < n.  Order of terms: for i in [0,lag): for l in {0,2lag,..,span-2lag}:
//   x[l+lag+i]*(x[l+i]+x[l+2lag+i]);  then the leftover products x[span+lag+i]*x[span+i].
// The (i,l) nest is flattened so that lanes with different lags share one loop.
__device__ __forceinline__ double chain_lag(const double* __restrict__ xs, uint32_t n, uint32_t lag)
{
  const uint32_t lag2 = lag << 1;
  const uint32_t groups = ((3 * lag) < n) ? (1 + (n - 3 * lag) / lag2) : 0;
  const uint32_t span = groups * lag2;
  double acc = 0.0;
  if (groups > 0) {
    // The walk is kept as ONE byte offset: a step moves it by 2*lag samples, the last step of an i by
    // 1 - (groups-1)*2*lag samples (back to the start, one sample further).  Per step that is a decrement, a compare,
    // two selects and three adds -- the chains are bound by VALU issue, so every integer instruction counts.
    const uint32_t steps = groups * lag;
    const uint32_t lagB = lag * 8u, lag2B = lag * 16u;
    const uint32_t wrapB = 8u - (groups - 1u) * lag2B;      // modulo 2^32: added to the offset
    const int32_t stepB = (int32_t)lag2B, backB = (int32_t)wrapB;
    const char* p = reinterpret_cast<const char*>(xs);
    uint32_t left = groups, s = 0;
    for (; s + CHAIN_U <= steps; s += CHAIN_U) {
      double a[CHAIN_U], c[CHAIN_U], b[CHAIN_U];
#pragma unroll
      for (int u = 0; u < CHAIN_U; u++) {
        a[u] = *reinterpret_cast<const double*>(p);
        c[u] = *reinterpret_cast<const double*>(p + lagB);
        b[u] = *reinterpret_cast<const double*>(p + lag2B);
        left--;
        const bool wrap = (left == 0);
        p += wrap ? backB : stepB;
        left = wrap ? groups : left;
      }
#pragma unroll
      for (int u = 0; u < CHAIN_U; u++) { acc += c[u] * (a[u] + b[u]); }
    }
    for (; s < steps; s++) {
      const double a = *reinterpret_cast<const double*>(p);
      const double c = *reinterpret_cast<const double*>(p + lagB);
      const double b = *reinterpret_cast<const double*>(p + lag2B);
      acc += c * (a + b);
      left--;
      const bool wrap = (left == 0);
      p += wrap ? backB : stepB;
      left = wrap ? groups : left;
    }
  }
  const uint32_t rest = n - span - lag;
  const double* t = xs + span;
  for (uint32_t k = 0; k < rest; k++) { acc += t[lag + k] * t[k]; }
  return acc;
}

// x86 cvttsd2si semantics of (int32_t)double (reference quantiser runs on x86-64)
__device__ __forceinline__ int32_t f64_to_i32_x86(double v)
{
  if (!(v > -2147483649.0 && v < 2147483648.0)) { return (int32_t)0x80000000; }
  return (int32_t)v;
}

// Levinson-Durbin in the reference's u/v formulation (src/SLAPredictor.c:253-328): o = { r0, parcor[0..order] }.
// a, v: order+2 doubles of work space each.
__device__ __forceinline__ void levinson_out(const double* rc, double* a, double* v, double* o, uint32_t order, uint32_t n)
{
  const uint32_t O1 = order + 1, O2 = order + 2;
  o[0] = rc[0];
  if (n < order || fabs(rc[0]) < (double)FLT_EPSILON) {
    for (uint32_t i = 0; i < O1; i++) { o[1 + i] = 0.0; }
  } else {
    for (uint32_t i = 0; i < O2; i++) { a[i] = 0.0; v[i] = 0.0; }
    a[0] = 1.0;
    a[1] = -rc[1] / rc[0];
    o[1] = 0.0;
    o[2] = rc[1] / rc[0];
    double e = rc[0] + rc[1] * a[1];
    for (uint32_t d = 1; d < order; d++) {
      double gamma = 0.0;
      for (uint32_t i = 0; i < d + 1; i++) { gamma += a[i] * rc[d + 1 - i]; }
      gamma /= (-e);
      e = (1.0 - gamma * gamma) * e;
      for (uint32_t i = 0; i < d; i++) { v[d - i] = a[i + 1]; }
      v[0] = 0.0; v[d + 1] = 1.0;
      a[0] = 1.0; a[d + 1] = 0.0;
      for (uint32_t i = 0; i < d + 2; i++) { a[i] = a[i] + gamma * v[i]; }
      o[2 + d] = -gamma;
    }
  }
}

// Final prediction error e_p = r0 * prod(1 - k_j^2) of the same recursion with r[0] replaced by r0 (no outputs);
// NaN as soon as the recursion leaves the positive-definite range.  Used by the search's certificate (k_search_finish).
__device__ __forceinline__ double levinson_error(const double* rc, double r0, double* a, double* v, uint32_t order)
{
  const double nan = __longlong_as_double(0x7FF8000000000000ll);
  if (!(r0 > 0.0)) { return nan; }
  for (uint32_t i = 0; i < order + 2; i++) { a[i] = 0.0; v[i] = 0.0; }
  a[0] = 1.0;
  a[1] = -rc[1] / r0;
  double e = r0 + rc[1] * a[1];
  if (!(e > 0.0)) { return nan; }
  for (uint32_t d = 1; d < order; d++) {
    double gamma = 0.0;
    for (uint32_t i = 0; i < d + 1; i++) { gamma += a[i] * rc[d + 1 - i]; }
    gamma /= (-e);
    if (!(fabs(gamma) < 1.0)) { return nan; }
    e = (1.0 - gamma * gamma) * e;
    for (uint32_t i = 0; i < d; i++) { v[d - i] = a[i + 1]; }
    v[0] = 0.0; v[d + 1] = 1.0;
    a[0] = 1.0; a[d + 1] = 0.0;
    for (uint32_t i = 0; i < d + 2; i++) { a[i] = a[i] + gamma * v[i]; }
  }
  return e;
}

#define LPC_MAX_PACK 4      // windows ("groups") one workgroup stages side by side

// A workgroup takes `pack` consecutive groups: their windows sit next to each other in LDS and their
// (candidate, lag) chains are numbered through, so that one wave-instruction of the chain loop carries
// up to 64 busy lanes even when a single group has only a few chains (chosen blocks: `order` chains).
// The chain loop is bound by FP64 / LDS issue per WAVE, not per lane, so lanes are what has to be filled.
__global__ __launch_bounds__(512)
void k_lpc(const int32_t* __restrict__ pcm, uint64_t stride, uint32_t ms, uint32_t order,
           const sla_hip_lpc_group* __restrict__ groups, uint32_t num_groups, uint32_t pack,
           const sla_hip_lpc_cand* __restrict__ cands,
           const double* __restrict__ window_pool, double* __restrict__ out,
           int32_t* __restrict__ out_code, int32_t* __restrict__ out_kint, uint32_t* __restrict__ out_rshift,
           uint32_t x_region, uint32_t mode, uint32_t* __restrict__ rerun_counter)
{
  extern __shared__ __attribute__((aligned(16))) double lds[];
  __shared__ sla_hip_lpc_group s_g[LPC_MAX_PACK];
  __shared__ uint32_t s_coff[LPC_MAX_PACK + 1];      // running candidate count
  __shared__ uint32_t s_maxabs[LPC_MAX_PACK];
  const uint32_t O1 = order + 1, O2 = order + 2;
  const uint32_t g0 = blockIdx.x * pack;
  const uint32_t ng = (num_groups - g0 < pack) ? (num_groups - g0) : pack;
  if (threadIdx.x < LPC_MAX_PACK) {
    if (threadIdx.x < ng) { s_g[threadIdx.x] = groups[g0 + threadIdx.x]; }
    s_maxabs[threadIdx.x] = 0;
  }
  __syncthreads();
  if (mode & 128u) {
    // rerun mode (after k_search_finish): only groups whose window was over the exactness limit -- their first
    // output slot carries the NaN flag -- are analysed, everything else keeps its tile-sum result
    uint32_t flagged = 0;
    for (uint32_t k = 0; k < ng; k++) { const double r0 = out[(uint64_t)s_g[k].slot_first * (order + 2)]; flagged += (r0 != r0) ? 1u : 0u; }
    if (flagged == 0) { return; }
    if (threadIdx.x == 0 && rerun_counter != nullptr) { atomicAdd(rerun_counter, flagged); }
  }
  if (threadIdx.x == 0) {
    uint32_t acc = 0;
    for (uint32_t k = 0; k < ng; k++) { s_coff[k] = acc; acc += s_g[k].cand_count; }
    for (uint32_t k = ng; k <= LPC_MAX_PACK; k++) { s_coff[k] = acc; }
  }
  __syncthreads();
  const uint32_t nc = s_coff[ng];                    // candidates of the whole pack
  double* r = lds + (size_t)pack * x_region;         // [nc*O1] after the windows
  double* av = lds;                                  // [nc*O2] overlay (windows are dead once every chain finished)
  double* vv = lds + (size_t)nc * O2;                // [nc*O2] overlay

  // ---- stage the windows (A0 + A4): convert, mid/side, window, pre-emphasis -------------------
  for (uint32_t k = 0; k < ng; k++) {
    const sla_hip_lpc_group g = s_g[k];
    double* x = lds + (size_t)k * x_region;
    const bool windowed = (g.win_off != SLA_HIP_NO_WINDOW);
    const double* win = window_pool + (windowed ? g.win_off : 0);
    uint32_t maxabs = 0;
    for (uint32_t s = threadIdx.x; s < g.num_samples; s += blockDim.x) {
      // mode bit 512 (per-call SLAPredictor API): `pcm` is an array of doubles the caller has already prepared
      double cur = (mode & 512u) ? reinterpret_cast<const double*>(pcm)[g.pcm_off + s]
                                     : load_f64(pcm, stride, ms, g.channel, g.pcm_off + s);
      if (windowed) {
        cur *= win[s];
        double prev = (s > 0) ? load_f64(pcm, stride, ms, g.channel, g.pcm_off + s - 1) * win[s - 1] : 0.0;
        cur -= prev * 0.96875;        // (2^5-1)*2^-5, src/SLAPredictor.c:1803-1809
      }
      x[s] = cur;
      if (out_code != nullptr) {
        int32_t v = load_int(pcm, stride, ms, g.channel, g.pcm_off + s, g.int_shift);
        uint32_t a = (v > 0) ? (uint32_t)v : (0u - (uint32_t)v);
        maxabs = (a > maxabs) ? a : maxabs;
      }
    }
    if (out_code != nullptr) {
      maxabs = umax_wave(maxabs);
      if ((threadIdx.x & 63) == 0) { atomicMax(&s_maxabs[k], maxabs); }
    }
  }
  __syncthreads();

  // pack-wide candidate index -> (group k, candidate entry)
  auto locate = [&](uint32_t cidx, uint32_t& k) -> const sla_hip_lpc_cand* {
    k = 0;
    while (k + 1 < ng && cidx >= s_coff[k + 1]) { k++; }
    return cands + s_g[k].cand_first + (cidx - s_coff[k]);
  };

  // ---- autocorrelation chains -----------------------------------------------------------------
  // waves 0..2 walk the lag >= 1 chains, wave 3 the lag-0 (energy) chains.  (Measured on C2: running the
  // two loop bodies back to back in the idle lanes of one wave is 1.1-1.4x slower than this split.)
  if (threadIdx.x < blockDim.x - 64) {
    const uint32_t nchains = nc * order;
    for (uint32_t q = threadIdx.x; q < nchains; q += blockDim.x - 64) {
      const uint32_t cidx = q / order, lag = 1 + (q - cidx * order);
      uint32_t k;
      const sla_hip_lpc_cand* cd = locate(cidx, k);
      const uint32_t n = cd->len;
      const double* xs = lds + (size_t)k * x_region + cd->start;
      r[cidx * O1 + lag] = (lag < n) ? chain_lag(xs, n, lag) : 0.0;
    }
  } else {
    for (uint32_t cidx = threadIdx.x - (blockDim.x - 64); cidx < nc; cidx += 64) {
      uint32_t k;
      const sla_hip_lpc_cand* cd = locate(cidx, k);
      r[cidx * O1] = chain_lag0(lds + (size_t)k * x_region + cd->start, cd->len);
    }
  }
  __syncthreads();

  // ---- Levinson-Durbin (+ quantiser for chosen blocks), one thread per candidate ----------------
  for (uint32_t cidx = threadIdx.x; cidx < nc; cidx += blockDim.x) {
    uint32_t k;
    const sla_hip_lpc_cand* cd = locate(cidx, k);
    const uint32_t n = cd->len;
    const uint64_t slot = (uint64_t)s_g[k].slot_first + (cidx - s_coff[k]);
    const double* rc = r + cidx * O1;
    double* a = av + cidx * O2;
    double* v = vv + cidx * O2;
    double* o = out + slot * O2;
    levinson_out(rc, a, v, o, order, n);
    // coefficient quantiser (chosen blocks: one candidate per group); the thread reads its own stores
    if (out_code != nullptr) {
      const uint32_t m = s_maxabs[k];
      // bit width = ceil(log2(max|x|)) + 1, at least 1                  src/SLAUtility.c:677-696
      const uint32_t l2c = (m > 1) ? (32u - (uint32_t)__builtin_clz(m - 1u)) : 0u;
      const uint32_t bitwidth = (m > 0) ? (l2c + 1u) : 1u;
      const uint32_t rshift = (bitwidth > 16) ? (bitwidth - 16) : 0;
      out_rshift[slot] = rshift; out_code[slot * O1] = 0; out_kint[slot * O1] = 0;
      for (uint32_t ord = 1; ord <= order; ord++) {
        const uint32_t q = (ord < 4) ? 16 : 8;
        const int32_t lim = 1 << (q - 1);
        double kq = o[1 + ord] * (double)lim;
        double rk = (kq >= 0.0) ? floor(kq + 0.5) : -floor(-kq + 0.5);
        int32_t code = f64_to_i32_x86(rk);
        code = (code < -lim) ? -lim : code;
        code = (code > lim - 1) ? (lim - 1) : code;
        out_code[slot * O1 + ord] = code;
        out_kint[slot * O1 + ord] = (int32_t)((uint32_t)code << (16u - q)) >> rshift;
      }
    }
  }
}

#define LAT_T 16
// (Measured and dropped: product and addition as ONE v_mad_u64_u32 whose high half is thrown away.  With two waves on a
// SIMD it issues every 2.0 ns against 3.3 ns for v_mul_lo_u32 + v_add_u32 (tests/tools/ubench_int.hip), but at the eight
// waves per SIMD this kernel runs with it takes the two issue slots of the pair: C5 16.1 ms per step either way.)
__device__ __forceinline__ int32_t lat_term(int32_t k, int32_t v)
{
  return (int32_t)((uint32_t)k * (uint32_t)v + 16384u) >> 15;
}

// One wave, `count` output samples of one (block, channel) starting at chunk_start (see k_lattice).  kc[1..order]:
// the block's lattice coefficients (wave-uniform reads).
__device__ __forceinline__ void lattice_chunk_wave(const int32_t* __restrict__ pcm, uint64_t stride, uint32_t ms, uint32_t order,
                                                   uint64_t blk_off, uint32_t blk_len, uint32_t chunk_start, uint32_t count,
                                                   uint32_t channel, uint32_t int_shift, const int32_t* __restrict__ kc,
                                                   int32_t* __restrict__ residual, uint32_t lane, bool raw = false)
{
  const uint32_t halo_lanes = (order + LAT_T - 1) / LAT_T;
  // sample index (relative to block) of this lane's first element; negative = before the block
  const int64_t first = (int64_t)chunk_start + ((int64_t)lane - (int64_t)halo_lanes) * LAT_T;
  int32_t f[LAT_T], b[LAT_T];
  // pre-emphasised input: y[n] = x[n] - ((x[n-1]*31)>>5), x[-1] = 0, zero outside the block
  int32_t prev = 0;
  {
    int64_t p = first - 1;
    if (p >= 0 && p < (int64_t)blk_len) { prev = load_int(pcm, stride, ms, channel, blk_off + p, int_shift); }
  }
#pragma unroll
  for (int i = 0; i < LAT_T; i++) {
    int64_t p = first + i;
    int32_t cur = 0;
    if (p >= 0 && p < (int64_t)blk_len) { cur = load_int(pcm, stride, ms, channel, blk_off + p, int_shift); }
    // raw: the caller's samples are the lattice input as they are (per-call SLALPCSynthesizer API)
    int32_t y = raw ? cur : (int32_t)((uint32_t)cur - (uint32_t)((int32_t)((uint32_t)prev * 31u) >> 5));
    f[i] = y; b[i] = y;
    prev = cur;
  }
  for (uint32_t m = 1; m <= order; m++) {
    const int32_t k = kc[m];                       // wave-uniform -> scalar load
    int32_t carry = __shfl_up(b[LAT_T - 1], 1);    // b_{m-1} of the sample just before this lane's run
    if (lane == 0) { carry = 0; }
#pragma unroll
    for (int i = LAT_T - 1; i >= 1; i--) {
      int32_t nf = (int32_t)((uint32_t)f[i] - (uint32_t)lat_term(k, b[i - 1]));
      int32_t nb = (int32_t)((uint32_t)b[i - 1] - (uint32_t)lat_term(k, f[i]));
      f[i] = nf; b[i] = nb;
    }
    {
      int32_t nf = (int32_t)((uint32_t)f[0] - (uint32_t)lat_term(k, carry));
      int32_t nb = (int32_t)((uint32_t)carry - (uint32_t)lat_term(k, f[0]));
      f[0] = nf; b[0] = nb;
    }
  }
  if (lane >= halo_lanes) {
    int32_t* dst = residual + (uint64_t)channel * stride + blk_off;
#pragma unroll
    for (int i = 0; i < LAT_T; i++) {
      int64_t p = first + i;
      if (p >= (int64_t)chunk_start && p < (int64_t)chunk_start + count) { dst[p] = f[i]; }
    }
  }
}

// The same wave, with its samples and its results crossing global memory coalesced.  A lane owns LAT_T = 16 consecutive
// samples, so when every lane loads and stores its own run, one access of the wave touches 64 different cache lines --
// 2048 line transactions per wave for 8 KB of traffic, as many cycles on the CU's one address path as the lattice takes
// on the wave's SIMD.  Here the wave reads its 1025 contiguous samples 64 at a time (4 transactions per access) into an
// LDS tile padded by one word per 16 (lane stride 17: conflict-free when the lanes then pick up their runs), and the
// residuals go back the same way.
//
// Round 4: the stages are written out instruction by instruction (the compiler's version of the C loop above carried 17
// register moves per stage, fetched every coefficient with a VECTOR load it then waited for, and needed 115 VGPRs), and
// each stage takes the shortest form its operands are PROVEN to allow.  The reference's term is
//     R(k, v) = (int32)(k * v + 2^14) >> 15        with the 32-bit product wrapping (src/SLAPredictor.c:590,596)
// and a stage is f[n] -= R(k, b[n-1]), b[n] = b[n-1] - R(k, f[n]).  The wave keeps a bound `bnd` >= every |f|, |b| it holds
// (the maximum of its own inputs, then bnd += ((|k| bnd + 2^14) >> 15) + 1 per stage) in a scalar register; with
// T = |k| bnd + 2^14 >= |k v + 2^14| a stage is, in this order of preference (cost = issue time per term in units of one
// v_add_u32 at eight waves per SIMD, tests/tools/ubench_lattice.hip: only add / sub / logic ops issue at 32 lanes per clock on
// gfx950, multiplies, shifts and SDWA forms at about half that):
//   H  T < 2^31 and |k| < 2^14:   the HIGH dword of v_mad_i64_i32(k << 17, v, 2^31) is (k v + 2^14) >> 15 without any wrap, which
//      is the reference's value when its own product does not wrap; then one subtraction: 2 instructions per term (3.3)
//   S  T < 2^30, bnd < 2^23:      v_mad_i32_i24(2k, v, 2^15), and the subtraction takes the sign-extended HIGH WORD of that as
//      its operand (SDWA): (2kv + 2^15) >> 16 = (kv + 2^14) >> 15 while nothing wraps: 2 per term (3.4); 16-bit material, |k| >= 2^14
//   M  bnd < 2^23:                v_mad_i32_i24(k, v, 2^14), >> 15, subtraction: 3 per term (4.9); the low 32 bits of the 48-bit
//      product are the wrapped product, so this form needs no statement about wrapping
//      (measured and dropped: v_mul_lo_u32(2k, v), + 2^15, SDWA subtraction for wide operands, 4.8 -- whenever it applies, H does)
//   W  otherwise:                 v_mul_lo_u32, + 2^14, >> 15, subtraction: 4 per term (6.4), the reference as it stands
// Nothing here is approximate: where the bound holds the forms are the same function of (k, v) as W, and where it does not
// hold W runs.  F[i] = f[n0 + i], B[i] = b[n0 + i - 1] (the backward error one sample late, as the stage wants it): a stage
// walks i downwards and writes the new backward error of sample n0 + i into B[i + 1], whose old value the step before has
// used up, so nothing is moved; B[0] arrives from the previous lane's B[16] by one DPP wave shift.  A result is never read
// by the instruction right behind its producer (two terms are in flight), so the blocks need no wait states.  The two
// products live in v[60:63] (inline asm cannot name the high half of a 64-bit operand).
#define LAT_TILE_WORDS 1104         // 1025 samples + one pad word per 16, rounded up
__device__ __forceinline__ uint32_t lat_pad(uint32_t p) { return p + (p >> 4); }

#define LAT_HI " dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_1\n"
#define LAT_SHIFT_IN "v_mov_b32_dpp %[B0], %[B16] wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:0\n"
// one sample of a full stage: B[j] = B[i] - R(k, F[i]), F[i] -= R(k, B[i])      (j = i + 1)
#define LAT_W(i, j) "v_mul_lo_u32 v60, %[k], %[F" #i "]\n v_mul_lo_u32 v62, %[k], %[B" #i "]\n v_add_u32 v60, 0x4000, v60\n v_add_u32 v62, 0x4000, v62\n" \
                    "v_ashrrev_i32 v60, 15, v60\n v_ashrrev_i32 v62, 15, v62\n v_sub_u32 %[B" #j "], %[B" #i "], v60\n v_sub_u32 %[F" #i "], %[F" #i "], v62\n"
#define LAT_M(i, j) "v_mad_i32_i24 v60, %[k], %[F" #i "], %[c]\n v_mad_i32_i24 v62, %[k], %[B" #i "], %[c]\n v_ashrrev_i32 v60, 15, v60\n v_ashrrev_i32 v62, 15, v62\n" \
                    "v_sub_u32 %[B" #j "], %[B" #i "], v60\n v_sub_u32 %[F" #i "], %[F" #i "], v62\n"
#define LAT_S(i, j) "v_mad_i32_i24 v60, %[k2], %[F" #i "], %[c2]\n v_mad_i32_i24 v62, %[k2], %[B" #i "], %[c2]\n" \
                    "v_sub_u32_sdwa %[B" #j "], %[B" #i "], sext(v60)" LAT_HI "v_sub_u32_sdwa %[F" #i "], %[F" #i "], sext(v62)" LAT_HI
#define LAT_H(i, j) "v_mad_i64_i32 v[60:61], vcc, %[k17], %[F" #i "], %[c31]\n v_mad_i64_i32 v[62:63], vcc, %[k17], %[B" #i "], %[c31]\n" \
                    "v_sub_u32 %[B" #j "], %[B" #i "], v61\n v_sub_u32 %[F" #i "], %[F" #i "], v63\n"
// two samples of the LAST stage, whose backward error has no reader: F[i] -= R(k, B[i]), F[j] -= R(k, B[j])
#define LAT_W1(i, j) "v_mul_lo_u32 v60, %[k], %[B" #i "]\n v_mul_lo_u32 v62, %[k], %[B" #j "]\n v_add_u32 v60, 0x4000, v60\n v_add_u32 v62, 0x4000, v62\n" \
                     "v_ashrrev_i32 v60, 15, v60\n v_ashrrev_i32 v62, 15, v62\n v_sub_u32 %[F" #i "], %[F" #i "], v60\n v_sub_u32 %[F" #j "], %[F" #j "], v62\n"
#define LAT_M1(i, j) "v_mad_i32_i24 v60, %[k], %[B" #i "], %[c]\n v_mad_i32_i24 v62, %[k], %[B" #j "], %[c]\n v_ashrrev_i32 v60, 15, v60\n v_ashrrev_i32 v62, 15, v62\n" \
                     "v_sub_u32 %[F" #i "], %[F" #i "], v60\n v_sub_u32 %[F" #j "], %[F" #j "], v62\n"
#define LAT_S1(i, j) "v_mad_i32_i24 v60, %[k2], %[B" #i "], %[c2]\n v_mad_i32_i24 v62, %[k2], %[B" #j "], %[c2]\n" \
                     "v_sub_u32_sdwa %[F" #i "], %[F" #i "], sext(v60)" LAT_HI "v_sub_u32_sdwa %[F" #j "], %[F" #j "], sext(v62)" LAT_HI
#define LAT_H1(i, j) "v_mad_i64_i32 v[60:61], vcc, %[k17], %[B" #i "], %[c31]\n v_mad_i64_i32 v[62:63], vcc, %[k17], %[B" #j "], %[c31]\n" \
                     "v_sub_u32 %[F" #i "], %[F" #i "], v61\n v_sub_u32 %[F" #j "], %[F" #j "], v63\n"
#define LAT_FULL(S) S(15, 16) S(14, 15) S(13, 14) S(12, 13) S(11, 12) S(10, 11) S(9, 10) S(8, 9) S(7, 8) S(6, 7) S(5, 6) S(4, 5) S(3, 4) S(2, 3) S(1, 2) S(0, 1)
#define LAT_HALF(S) S(15, 14) S(13, 12) S(11, 10) S(9, 8) S(7, 6) S(5, 4) S(3, 2) S(1, 0)
// ONE asm statement per stage, the form chosen by scalar branches inside it: with one statement per form the register
// allocator gave each its own assignment of the 33 state registers and moved (and spilled) them in front of every stage
#define LAT_DISPATCH(W, M, S, H) LAT_SHIFT_IN \
  "s_cmp_eq_u32 %[form], 3\n s_cbranch_scc0 .Llat_nh%=\n" \
  H "s_branch .Llat_end%=\n" \
  ".Llat_nh%=:\n s_cmp_eq_u32 %[form], 2\n s_cbranch_scc0 .Llat_wm%=\n" \
  S "s_branch .Llat_end%=\n" \
  ".Llat_wm%=:\n s_cmp_eq_u32 %[form], 1\n s_cbranch_scc1 .Llat_m%=\n" \
  W "s_branch .Llat_end%=\n" \
  ".Llat_m%=:\n" M \
  ".Llat_end%=:\n"
#define LAT_REGS [F0] "+v"(F[0]), [F1] "+v"(F[1]), [F2] "+v"(F[2]), [F3] "+v"(F[3]), [F4] "+v"(F[4]), [F5] "+v"(F[5]), [F6] "+v"(F[6]), [F7] "+v"(F[7]), \
                 [F8] "+v"(F[8]), [F9] "+v"(F[9]), [F10] "+v"(F[10]), [F11] "+v"(F[11]), [F12] "+v"(F[12]), [F13] "+v"(F[13]), [F14] "+v"(F[14]), [F15] "+v"(F[15]), \
                 [B0] "+v"(B[0]), [B1] "+v"(B[1]), [B2] "+v"(B[2]), [B3] "+v"(B[3]), [B4] "+v"(B[4]), [B5] "+v"(B[5]), [B6] "+v"(B[6]), [B7] "+v"(B[7]), [B8] "+v"(B[8]), \
                 [B9] "+v"(B[9]), [B10] "+v"(B[10]), [B11] "+v"(B[11]), [B12] "+v"(B[12]), [B13] "+v"(B[13]), [B14] "+v"(B[14]), [B15] "+v"(B[15]), [B16] "+v"(B[16])
#define LAT_INS [k] "v"(k), [k2] "v"(k * 2), [k17] "v"((int32_t)((uint32_t)k << 17)), [c] "s"(0x4000), [c2] "s"(0x8000), [c31] "s"(0x80000000ll), [form] "s"(form)
#define LAT_CLOBBERS "scc", "vcc", "v60", "v61", "v62", "v63"
static_assert(LAT_T == 16, "the stage blocks are written for 16 samples per lane");

// the form of one stage and the bound behind it (all wave-uniform: scalar registers)
enum { LAT_FORM_W = 0, LAT_FORM_M = 1, LAT_FORM_S = 2, LAT_FORM_H = 3 };
__device__ __forceinline__ int lat_pick_form(int32_t k, uint32_t& bnd, bool plain)
{
  // T = |k| bnd + 2^14 >= |k v + 2^14| for every operand this wave holds; kept as (hi, lo) of the 64-bit product so that every
  // comparison is a 32-bit scalar one
  const uint32_t ak = (k < 0) ? (0u - (uint32_t)k) : (uint32_t)k;
  const uint32_t lo = ak * bnd, hi = __umulhi(ak, bnd);
  const bool t31 = (hi == 0u) && (lo < 0x80000000u - 16384u);        // T < 2^31: the reference's own product does not wrap
  const bool t30 = (hi == 0u) && (lo < 0x40000000u - 16384u);        // T < 2^30: 2 (k v + 2^14) is an int32
  const bool high = t31 && (ak < (1u << 14));                        // k << 17 is an int32
  const bool v24 = (bnd < (1u << 23)) && (ak < (1u << 22));          // v, k and 2k are 24-bit signed values
  const int form = plain ? LAT_FORM_W : high ? LAT_FORM_H : !v24 ? LAT_FORM_W : t30 ? LAT_FORM_S : LAT_FORM_M;
  // |R| <= (|k v| + 2^14) >> 15 rounded up where nothing wraps; a wrapped product still gives |R| <= 2^16
  const uint32_t grow = t31 ? (((lo + 16384u) >> 15) + 1u) : 65537u;
  const uint32_t nb = bnd + grow;                                     // bnd <= 2^31, grow <= 2^16 + 1: no overflow
  bnd = (nb < (1u << 31)) ? nb : (1u << 31);
  return __builtin_amdgcn_readfirstlane(form);
}

__device__ __forceinline__ void lattice_chunk_wave_lds(const int32_t* __restrict__ pcm, uint64_t stride, uint32_t ms, uint32_t order,
                                                       uint64_t blk_off, uint32_t blk_len, uint32_t chunk_start, uint32_t count,
                                                       uint32_t channel, uint32_t int_shift, const int32_t* __restrict__ kc,
                                                       int32_t* __restrict__ residual, uint32_t lane, bool raw, int32_t* __restrict__ tile,
                                                       bool plain)
{
  const uint32_t halo_lanes = (order + LAT_T - 1) / LAT_T;
  const int32_t base = (int32_t)chunk_start - (int32_t)(halo_lanes * LAT_T);    // block-relative index of lane 0's first sample (blocks are < 2^31 samples)
  // tile slot e holds sample base - 1 + e (e = 0 .. 1024), zero outside the block.  Every load is requested before the first
  // one is looked at (the first version's seventeen guarded loads each waited for its own data), from a clamped in-block
  // address, and a sample outside the block is zeroed afterwards: no branches, offsets of 32 bits from a scalar base
  {
    const int32_t* __restrict__ src = pcm + (ms ? (uint64_t)0 : (uint64_t)channel * stride) + blk_off;
    int32_t va[17];
    bool ok[17];
    uint32_t pc[17];
#pragma unroll
    for (int j = 0; j < 17; j++) {
      const uint32_t e = (uint32_t)j * 64u + lane;
      const uint32_t p = (uint32_t)(base - 1) + e;                          // a sample before the block wraps to a huge index
      ok[j] = (p < blk_len) && (e <= (uint32_t)(SLA_WAVE * LAT_T));
      pc[j] = ok[j] ? p : 0u;
    }
    if (!ms) {
#pragma unroll
      for (int j = 0; j < 17; j++) { va[j] = src[pc[j]]; }
#pragma unroll
      for (int j = 0; j < 17; j++) { va[j] = ok[j] ? (va[j] >> int_shift) : 0; }
    } else {
      // mid = (L+R)>>1 (arithmetic, wrapping sum), side = L-R      src/SLAUtility.c:403-411
      const int32_t* __restrict__ srcr = src + stride;
      int32_t vr[17];
#pragma unroll
      for (int j = 0; j < 17; j++) { va[j] = src[pc[j]]; vr[j] = srcr[pc[j]]; }
#pragma unroll
      for (int j = 0; j < 17; j++) {
        const int32_t l = va[j] >> int_shift, r = vr[j] >> int_shift;
        const int32_t v = (channel == 0) ? ((int32_t)((uint32_t)l + (uint32_t)r) >> 1) : (int32_t)((uint32_t)l - (uint32_t)r);
        va[j] = ok[j] ? v : 0;
      }
    }
#pragma unroll
    for (int j = 0; j < 17; j++) {
      const uint32_t e = (uint32_t)j * 64u + lane;
      if (j < 16 || e <= (uint32_t)(SLA_WAVE * LAT_T)) { tile[lat_pad(e)] = va[j]; }
    }
  }
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  int32_t F[LAT_T], B[LAT_T + 1];
  int32_t hi = 0, lo = 0;
  {
    int32_t prev = tile[lat_pad(lane * LAT_T)];
#pragma unroll
    for (int i = 0; i < LAT_T; i++) {
      const int32_t cur = tile[lat_pad(lane * LAT_T + 1 + (uint32_t)i)];
      // pre-emphasis y[n] = x[n] - ((x[n-1]*31)>>5); raw: the caller's samples are the lattice input as they are
      const int32_t y = raw ? cur : (int32_t)((uint32_t)cur - (uint32_t)((int32_t)((uint32_t)prev * 31u) >> 5));
      F[i] = y; B[i + 1] = y;
      hi = max(hi, y); lo = min(lo, y);
      prev = cur;
    }
  }
  B[0] = 0;
  // bnd >= |y| of every sample this wave holds (its halo included): max(hi, -lo) as unsigned covers INT32_MIN
  uint32_t bnd = (uint32_t)__builtin_amdgcn_readfirstlane((int)umax_wave(max((uint32_t)hi, 0u - (uint32_t)lo)));
  int32_t knext = kc[1];
  for (uint32_t m = 1; m < order; m++) {
    const int32_t k = __builtin_amdgcn_readfirstlane(knext);
    knext = kc[m + 1];                               // (the last full stage fetches the last stage's coefficient)
    const int form = lat_pick_form(k, bnd, plain);
    asm volatile(LAT_DISPATCH(LAT_FULL(LAT_W), LAT_FULL(LAT_M), LAT_FULL(LAT_S), LAT_FULL(LAT_H)) : LAT_REGS : LAT_INS : LAT_CLOBBERS);
  }
  if (order >= 1) {
    // the last stage: only the forward error leaves the lattice, its backward error has no reader (half the stage's work)
    const int32_t k = __builtin_amdgcn_readfirstlane(knext);
    const int form = lat_pick_form(k, bnd, plain);
    asm volatile(LAT_DISPATCH(LAT_HALF(LAT_W1), LAT_HALF(LAT_M1), LAT_HALF(LAT_S1), LAT_HALF(LAT_H1)) : LAT_REGS : LAT_INS : LAT_CLOBBERS);
  }
  // results through the tile (slot e = sample base + e now), stored 64 consecutive samples at a time
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
  for (int i = 0; i < LAT_T; i++) { tile[lat_pad(lane * LAT_T + (uint32_t)i)] = F[i]; }
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  int32_t* __restrict__ dst = residual + (uint64_t)channel * stride + blk_off + chunk_start;
  int32_t out[LAT_T];
#pragma unroll
  for (int j = 0; j < LAT_T; j++) { out[j] = tile[lat_pad((uint32_t)j * 64u + lane)]; }
#pragma unroll
  for (int j = 0; j < LAT_T; j++) {
    const uint32_t q = (uint32_t)j * 64u + lane - halo_lanes * LAT_T;       // chunk-relative index; the halo wraps to a huge one
    if (q < count) { dst[q] = out[j]; }
  }
}

// ---------------------------------------------------------------------------------------------
// k_lpc_blocks: the chosen blocks (one candidate per group = the whole windowed block).
//
// A serial chain only has to be serial in its ADDITIONS: the terms c*(a+b) are independent.  So the work
// is split by role (512 threads).  Waves 2-7 ("producers") compute LB_K consecutive terms of every chain
// of the pack into an LDS tile, in the chain's own order (pairs i-major, then the leftover products; 0.0
// past the end, which leaves an accumulator untouched bit for bit); wave 0 owns one accumulator per
// (window, lag) and adds the previous tile -- half a ds_read_b128 and one v_add_f64 per step for up to 64
// chains at once; wave 1 does the same for the energy sums (two squares per step).  One barrier per tile,
// two tile buffers.  Measured on C2 (shader clock per workgroup of two 4096-sample windows): 265k ticks for
// the chains in k_lpc's one-lane-per-chain loop, 140k here; staging 62k -> 20k (straight-line batches);
// Levinson-Durbin + quantiser 60k -> 16k: it runs lane-parallel, one wave per window, lane j holds a[j]; the
// dot product is multiplied in parallel and summed in the reference's order through v_readlane; the
// reversed vector is a ds_bpermute.  The quantiser runs one lane per coefficient.
// LDS: x[pack][x_region] | terms[2][nch][LB_K+2] | sq[2][pack][2*LB_K] | r[pack][order+1]
// ---------------------------------------------------------------------------------------------
#define LB_K 24
#define LB_THREADS 512          // wave 0: lag accumulators, wave 1: energy accumulators, waves 2-7: term producers
#define LB_PRODUCERS (LB_THREADS - 128)
__device__ unsigned long long g_lpc_clk[8];   // SLA_HIP_LPC_CLK=1: shader-clock ticks per phase, summed over workgroups

__device__ __forceinline__ double readlane_f64(double v, int lane)      // lane must be wave-uniform
{
  return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), lane), __builtin_amdgcn_readlane(__double2loint(v), lane));
}

// Stage one analysis window into LDS: x[s] = w[s]*in[s] - 0.96875 * w[s-1]*in[s-1] (src/SLAEncoder.c:505-515,
// 540-543, src/SLAPredictor.c:1803-1809) and return the wave's max |integer sample| (for the quantiser's
// shift).  MS is a template parameter so that the batch below is straight-line code: all its samples are
// requested before the first one is used (one memory latency per batch of 8, not per sample).
template <bool MS>
__device__ __forceinline__ uint32_t stage_window(const int32_t* __restrict__ pcm, uint64_t stride, const sla_hip_lpc_group& g,
                                                 const double* __restrict__ win, double* __restrict__ x, uint32_t tid)
{
  const double scale = 4.656612873077392578125e-10;   // 2^-31, exact
  const int32_t* p0 = pcm + (MS ? 0 : (uint64_t)g.channel * stride) + g.pcm_off;
  const int32_t* p1 = pcm + stride + g.pcm_off;        // right channel (MS only)
  uint32_t maxabs = 0;
  for (uint32_t base = 0; base < g.num_samples; base += LB_THREADS * 8) {
    int32_t l0[8], l1[8], r0[8], r1[8];
    double w0[8], w1[8];
#pragma unroll
    for (int u = 0; u < 8; u++) {
      const uint32_t s = base + u * LB_THREADS + tid;
      const bool in = (s < g.num_samples);
      const uint32_t sc = in ? s : 0, sp = (in && s > 0) ? (s - 1) : 0;
      l0[u] = p0[sc]; l1[u] = p0[sp];
      if (MS) { r0[u] = p1[sc]; r1[u] = p1[sp]; }
      w0[u] = win[sc]; w1[u] = win[sp];
    }
#pragma unroll
    for (int u = 0; u < 8; u++) {
      const uint32_t s = base + u * LB_THREADS + tid;
      if (s < g.num_samples) {
        double cur, prev;
        int32_t iv;
        if (MS) {
          // mid = (l+r)/2, side = l-r in double (src/SLAUtility.c:382-387); integers: (L+R)>>1, L-R on the shifted samples (:403-411)
          const double a = (double)l0[u] * scale, b = (double)r0[u] * scale, c = (double)l1[u] * scale, d = (double)r1[u] * scale;
          cur = (g.channel == 0) ? ((a + b) / 2) : (a - b);
          prev = (g.channel == 0) ? ((c + d) / 2) : (c - d);
          const int32_t li = l0[u] >> g.int_shift, ri = r0[u] >> g.int_shift;
          iv = (g.channel == 0) ? ((int32_t)((uint32_t)li + (uint32_t)ri) >> 1) : (int32_t)((uint32_t)li - (uint32_t)ri);
        } else {
          cur = (double)l0[u] * scale; prev = (double)l1[u] * scale;
          iv = l0[u] >> g.int_shift;
        }
        const double c = cur * w0[u];
        const double pv = (s > 0) ? (prev * w1[u]) : 0.0;
        x[s] = c - pv * 0.96875;
        const uint32_t a = (iv > 0) ? (uint32_t)iv : (0u - (uint32_t)iv);
        maxabs = (a > maxabs) ? a : maxabs;
      }
    }
  }
  return umax_wave(maxabs);
}

#define LB_ROW (LB_K + 2)     // terms of one chain and tile, padded: 16-byte aligned rows that spread over the banks

template <int S, int LBK>     // producer lanes per chain: 6 or 12; steps per tile: 24 or 48 (longer tiles pay for wide packs: fewer barriers, more operand reuse)
__device__ __forceinline__
void lpc_blocks_body(const int32_t* __restrict__ pcm, uint64_t stride, uint32_t ms, uint32_t order,
                  const sla_hip_lpc_group* __restrict__ groups, uint32_t num_groups, uint32_t pack,
                  const double* __restrict__ window_pool, double* __restrict__ out,
                  int32_t* __restrict__ out_code, int32_t* __restrict__ out_kint, uint32_t* __restrict__ out_rshift,
                  uint32_t x_region, uint32_t nch, uint32_t clk,
                  int32_t* __restrict__ lat_residual, uint32_t defer_levinson, uint32_t g0, const uint32_t* __restrict__ list)
{
  constexpr uint32_t Q = LBK / S;              // consecutive terms a producer lane makes per tile
  extern __shared__ __attribute__((aligned(16))) double lds[];
  const unsigned long long t_start = clk ? clock64() : 0;
  __shared__ sla_hip_lpc_group s_g[LPC_MAX_PACK];
  __shared__ uint32_t s_maxabs[LPC_MAX_PACK];
  __shared__ uint32_t s_tiles;
  const uint32_t O1 = order + 1, O2 = order + 2;
  const uint32_t tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const uint32_t ng = (num_groups - g0 < pack) ? (num_groups - g0) : pack;
  if (tid < LPC_MAX_PACK) {
    if (tid < ng) { s_g[tid] = groups[(list != nullptr) ? list[g0 + tid] : (g0 + tid)]; }
    s_maxabs[tid] = 0;
  }
  if (tid == 0) { s_tiles = 0; }
  __syncthreads();
  double* terms = lds + (size_t)pack * x_region;            // [2][nch][(LBK + 2)]
  double* sq = terms + (size_t)2 * nch * (LBK + 2);            // [2][pack][2*LBK]
  double* r = sq + (size_t)2 * 2 * LBK * pack;             // [pack][O1]

  // ---- stage the windows: convert, mid/side, window, pre-emphasis (as k_lpc) -------------------------
  for (uint32_t k = 0; k < ng; k++) {
    const sla_hip_lpc_group g = s_g[k];
    const uint32_t maxabs = ms ? stage_window<true>(pcm, stride, g, window_pool + g.win_off, lds + (size_t)k * x_region, tid)
                               : stage_window<false>(pcm, stride, g, window_pool + g.win_off, lds + (size_t)k * x_region, tid);
    if (lane == 0) { atomicMax(&s_maxabs[k], maxabs); }
  }

  // ---- chain geometry: the consumer lane and the producer lanes of a chain derive the same numbers ------
  const uint32_t nchains = ng * order;
  const bool producer = (wv >= 2);
  const uint32_t pl = tid - 128;                            // producer lane number (valid when producer)
  const uint32_t ch = (wv == 0) ? lane : (producer ? (pl % nch) : 0xFFFFFFFFu);
  const uint32_t j = producer ? (pl / nch) : 0;             // which Q-term piece of the tile this producer lane makes
  const bool has_chain = (ch < nchains) && (!producer || j < (uint32_t)S);
  const uint32_t ck = has_chain ? (ch / order) : 0;
  const uint32_t lag = has_chain ? (1 + ch - ck * order) : 1;
  const uint32_t n = s_g[ck].num_samples;
  const uint32_t lag2 = lag << 1;
  const uint32_t grp = (has_chain && 3 * lag < n) ? (1 + (n - 3 * lag) / lag2) : 0;
  const uint32_t span = grp * lag2;
  const uint32_t npair = grp * lag;
  const uint32_t total = (has_chain && lag < n) ? (npair + (n - span - lag)) : 0;     // terms of this chain
  const double* xs = lds + (size_t)ck * x_region;
  {
    uint32_t need = (total + LBK - 1) / LBK;
    if (tid < ng) { const uint32_t e = (s_g[tid].num_samples + 2 * LBK - 1) / (2 * LBK); need = (e > need) ? e : need; }
    need = umax_wave(need);
    if (lane == 0) { atomicMax(&s_tiles, need); }
  }
  __syncthreads();
  const uint32_t ntiles = s_tiles;
  const unsigned long long t_staged = clk ? clock64() : 0;
  unsigned long long t_prod = 0, t_cons = 0;

  // producer position: term kk = tile*LBK + j*Q sits at (run pi, step pg) while it is a pair term
  uint32_t kk = j * Q;
  uint32_t pi = (grp != 0) ? (kk / grp) : 0, pg = (grp != 0) ? (kk - (kk / grp) * grp) : 0;
  uint32_t ppos = pi + pg * lag2;                           // sample index of a of that term
  // energy producer: lane e < 2*LBK*ng makes the square of sample tile*2*LBK + es of window ew
  const uint32_t nsq = 2 * LBK * ng;
  const uint32_t ew = pl / (2 * LBK), es = pl - ew * (2 * LBK);      // 2*LBK*LPC_MAX_PACK = 192 <= LB_PRODUCERS

  double acc = 0.0;
  for (uint32_t t = 0; t <= ntiles; t++) {
    const unsigned long long t_a = clk ? clock64() : 0;
    if (producer && t < ntiles) {
      // most tiles: every lane of this wave makes Q consecutive pair terms of one run -- no bookkeeping, and
      // b of one term is a of the next
      const bool plain = !has_chain || (kk + Q <= npair && pg + Q <= grp);
      if (__all(plain)) {
        if (has_chain) {
          double* dst = terms + ((size_t)(t & 1) * nch + ch) * (LBK + 2) + j * Q;
          const double* p = xs + ppos;
          double va[Q + 1], vc[Q];
#pragma unroll
          for (int q = 0; q < (int)Q; q++) { va[q] = p[q * lag2]; vc[q] = p[q * lag2 + lag]; }
          va[Q] = p[Q * lag2];
#pragma unroll
          for (int q = 0; q < (int)Q; q++) { dst[q] = vc[q] * (va[q] + va[q + 1]); }
        }
      } else if (has_chain) {
        double* dst = terms + ((size_t)(t & 1) * nch + ch) * (LBK + 2) + j * Q;
        uint32_t ii = pi, gg = pg, pos = ppos;
        uint32_t ia[Q]; bool pr[Q], vl[Q];
#pragma unroll
        for (int q = 0; q < (int)Q; q++) {
          const uint32_t kq = kk + q;
          pr[q] = (kq < npair); vl[q] = (kq < total);
          ia[q] = pr[q] ? pos : (vl[q] ? (span + (kq - npair)) : 0u);
          const bool wrap = (gg + 1 == grp);
          gg = wrap ? 0u : (gg + 1);
          ii += wrap ? 1u : 0u;
          pos = wrap ? ii : (pos + lag2);
        }
        double va[Q], vc[Q], vb[Q];
#pragma unroll
        for (int q = 0; q < (int)Q; q++) { va[q] = xs[ia[q]]; vc[q] = xs[ia[q] + lag]; vb[q] = xs[ia[q] + lag2]; }
#pragma unroll
        for (int q = 0; q < (int)Q; q++) {
          const double tv = vc[q] * (va[q] + (pr[q] ? vb[q] : 0.0));      // leftover products: c*a (a+0.0 == a up to the sign of zero, which a sum that starts at +0.0 cannot see)
          dst[q] = vl[q] ? tv : 0.0;
        }
      }
      if (has_chain) {
        kk += LBK;
        if (grp != 0) { pg += LBK; ppos += LBK * lag2; while (pg >= grp) { pg -= grp; pi++; ppos += 1 - span; } }
      }
      double* dq = sq + (size_t)(t & 1) * 2 * LBK * pack;
      if (pl < nsq) {
        const uint32_t idx = t * 2 * LBK + es;
        const double v = (idx < s_g[ew].num_samples) ? lds[(size_t)ew * x_region + idx] : 0.0;
        dq[ew * 2 * LBK + es] = v * v;
      }
    }
    const unsigned long long t_b = clk ? clock64() : 0;
    if (t >= 1) {
      if (wv == 0 && has_chain) {
        const double2* src = (const double2*)(terms + ((size_t)((t - 1) & 1) * nch + ch) * (LBK + 2));
        double2 v[LBK / 2];
#pragma unroll
        for (int q = 0; q < LBK / 2; q++) { v[q] = src[q]; }
        __builtin_amdgcn_sched_barrier(0);          // every load of the tile is in flight before the first add waits
#pragma unroll
        for (int q = 0; q < LBK / 2; q++) { acc += v[q].x; acc += v[q].y; }
      } else if (wv == 1 && lane < ng) {
        const double2* src = (const double2*)(sq + (size_t)((t - 1) & 1) * 2 * LBK * pack + lane * 2 * LBK);
        double2 v[LBK];
#pragma unroll
        for (int q = 0; q < LBK; q++) { v[q] = src[q]; }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int q = 0; q < LBK; q++) { acc += v[q].x; acc += v[q].y; }
      }
    }
    if (clk) { const unsigned long long t_c = clock64(); t_prod += t_b - t_a; t_cons += t_c - t_b; }
    __syncthreads();
  }
  if (wv == 0 && has_chain) { r[ck * O1 + lag] = acc; }
  if (wv == 1 && lane < ng) { r[lane * O1] = acc; }
  __syncthreads();
  const unsigned long long t_chained = clk ? clock64() : 0;

  if (defer_levinson) {
    // the recursion and the quantiser run in k_blocks_finish (one LANE per window, everything in registers): this
    // workgroup only hands over r[0..order] (in the slot's parcor area) and the quantiser's shift
    if (wv < ng) {
      const sla_hip_lpc_group g = s_g[wv];
      double* o = out + (uint64_t)g.slot_first * O2;
      if (lane <= order) { o[1 + lane] = r[wv * O1 + lane]; }
      if (lane == 0) {
        const uint32_t m = s_maxabs[wv];
        const uint32_t l2c = (m > 1) ? (32u - (uint32_t)__builtin_clz(m - 1u)) : 0u;    // src/SLAUtility.c:677-696
        const uint32_t bitwidth = (m > 0) ? (l2c + 1u) : 1u;
        out_rshift[g.slot_first] = (bitwidth > 16) ? (bitwidth - 16) : 0;
      }
    }
    return;
  }
  // ---- Levinson-Durbin, one wave per window, lane j = coefficient j   src/SLAPredictor.c:253-328 ------
  if (wv < ng) {
    const sla_hip_lpc_group g = s_g[wv];
    const uint64_t slot = g.slot_first;
    double* o = out + slot * O2;
    const double rc = (lane <= order) ? r[wv * O1 + lane] : 0.0;
    const double r0 = readlane_f64(rc, 0), r1 = readlane_f64(rc, 1);
    double par = 0.0;                               // lane j: parcor[j]
    if (!(g.num_samples < order || fabs(r0) < (double)FLT_EPSILON)) {
      const double a1 = -r1 / r0;
      double a = (lane == 0) ? 1.0 : ((lane == 1) ? a1 : 0.0);
      par = (lane == 1) ? (r1 / r0) : 0.0;
      double e = r0 + r1 * a1;
      for (uint32_t d = 1; d < order; d++) {
        const int src = (int)(d + 1) - (int)lane;                   // lanes 0..d read r[d+1-i] / a[d+1-i]
        const double rrev = __shfl(rc, src & 63);
        const double prod = a * rrev;
        double gamma = 0.0;
        for (uint32_t i = 0; i <= d; i++) { gamma += readlane_f64(prod, (int)i); }
        gamma /= (-e);
        e = (1.0 - gamma * gamma) * e;
        const double arev = __shfl(a, src & 63);
        const double v = (lane == 0) ? 0.0 : ((lane == d + 1) ? 1.0 : ((lane <= d) ? arev : 0.0));
        const double a_old = (lane == d + 1) ? 0.0 : a;
        a = a_old + gamma * v;
        par = (lane == d + 1) ? (-gamma) : par;
      }
    }
    if (lane <= order) { o[1 + lane] = par; }
    if (lane == 0) { o[0] = r0; }
    // coefficient quantiser, lane = coefficient                          src/SLAEncoder.c:567-589
    const uint32_t m = s_maxabs[wv];
    const uint32_t l2c = (m > 1) ? (32u - (uint32_t)__builtin_clz(m - 1u)) : 0u;    // src/SLAUtility.c:677-696
    const uint32_t bitwidth = (m > 0) ? (l2c + 1u) : 1u;
    const uint32_t rshift = (bitwidth > 16) ? (bitwidth - 16) : 0;
    if (lane == 0) { out_rshift[slot] = rshift; out_code[slot * O1] = 0; out_kint[slot * O1] = 0; }
    if (lane >= 1 && lane <= order) {
      const uint32_t qb = (lane < 4) ? 16 : 8;
      const int32_t lim = 1 << (qb - 1);
      const double kq = par * (double)lim;
      const double rk = (kq >= 0.0) ? floor(kq + 0.5) : -floor(-kq + 0.5);
      int32_t code = f64_to_i32_x86(rk);
      code = (code < -lim) ? -lim : code;
      code = (code > lim - 1) ? (lim - 1) : code;
      out_code[slot * O1 + lane] = code;
      out_kint[slot * O1 + lane] = (int32_t)((uint32_t)code << (16u - qb)) >> rshift;
    }
  }
  // ---- PARCOR lattice of the same blocks (k_lattice's wave-chunks), while their samples are still hot in L2 ----
  if (lat_residual != nullptr) {
    __threadfence();                                  // out_kint of this workgroup's slots: visible to its other waves
    __syncthreads();
    const uint32_t per = (SLA_WAVE - (order + LAT_T - 1) / LAT_T) * LAT_T;
    uint32_t first_chunk = 0;
    for (uint32_t k = 0; k < ng; k++) {
      const sla_hip_lpc_group g = s_g[k];
      const uint32_t nchunks = (g.num_samples + per - 1) / per;
      for (uint32_t cid = wv; cid < first_chunk + nchunks; cid += LB_THREADS / 64) {
        if (cid < first_chunk) { continue; }
        const uint32_t at = (cid - first_chunk) * per;
        lattice_chunk_wave(pcm, stride, ms, order, g.pcm_off, g.num_samples, at, (g.num_samples - at < per) ? (g.num_samples - at) : per,
                           g.channel, g.int_shift, out_kint + (uint64_t)g.slot_first * O1, lat_residual, lane);
      }
      first_chunk += nchunks;
    }
  }
  if (clk && lane == 0) {
    const unsigned long long t_end = clock64();
    if (wv == 0) {
      atomicAdd(&g_lpc_clk[0], t_staged - t_start); atomicAdd(&g_lpc_clk[1], t_chained - t_staged);
      atomicAdd(&g_lpc_clk[2], t_end - t_chained); atomicAdd(&g_lpc_clk[3], 1ull); atomicAdd(&g_lpc_clk[4], t_cons);
    }
    if (wv == 2) { atomicAdd(&g_lpc_clk[5], t_prod); }
    if (wv == 1) { atomicAdd(&g_lpc_clk[6], t_cons); }
  }
}

// list mode (list != NULL): the groups are list[0 .. *list_count) -- the blocks k_blocks_finish<.., true> could not
// certify -- and a fixed grid walks them (the count is only known on the device).
template <int S, int LBK>
__global__ __launch_bounds__(LB_THREADS)
void k_lpc_blocks(const int32_t* __restrict__ pcm, uint64_t stride, uint32_t ms, uint32_t order,
                  const sla_hip_lpc_group* __restrict__ groups, uint32_t num_groups, uint32_t pack,
                  const double* __restrict__ window_pool, double* __restrict__ out,
                  int32_t* __restrict__ out_code, int32_t* __restrict__ out_kint, uint32_t* __restrict__ out_rshift,
                  uint32_t x_region, uint32_t nch, uint32_t clk, unsigned long long* exec_span,
                  int32_t* __restrict__ lat_residual, uint32_t defer_levinson,
                  const uint32_t* __restrict__ list, const uint32_t* __restrict__ list_count)
{
  span_begin(exec_span);
  const uint32_t total = (list != nullptr) ? *list_count : num_groups;
  for (uint32_t g0 = blockIdx.x * pack; g0 < total; g0 += gridDim.x * pack) {
    lpc_blocks_body<S, LBK>(pcm, stride, ms, order, groups, total, pack, window_pool, out, out_code, out_kint, out_rshift,
                            x_region, nch, clk, lat_residual, defer_levinson, g0, list);
    __syncthreads();
  }
  span_end(exec_span);
}

// ---------------------------------------------------------------------------------------------
// k_blocks_finish: Levinson-Durbin (src/SLAPredictor.c:253-328) and the coefficient quantiser (src/SLAEncoder.c:567-589)
// of the chosen blocks, one LANE per (block, channel).  Inside k_lpc_blocks the recursion ran lane-parallel on one wave
// per window -- the sum gamma = sum a[i]*r[d+1-i] has to be added up in the reference's order, so it was (order^2)/2
// dependent v_readlane + v_add_f64 pairs while the other waves of the workgroup idled: 91 of 376 thousand cycles per
// C5 workgroup.  A lane that keeps a[] and r[] in registers does the same arithmetic, operation for operation, with no
// cross-lane traffic at all, and 64 windows share a wave; the loops are unrolled over the stage and the coefficient
// index so that every register index is static (the reversed vector v[i] = a[d+1-i] is just another register).
// In: slot = { -, r[0..order] } as k_lpc_blocks left it, rshift.  Out: slot = { r0, parcor[0..order] }, code, kint.
//
// CERT = true: r[] came from k_acf_blocks (any summation order), so the doubles are NOT the reference's bit for bit --
// but what reaches the bit stream are the quantised codes (16 bits for the first three coefficients, 8 bits for the
// rest, src/SLAEncoder.c:573-589) and the RAW decision (src/SLAEncoder.c:553-565).  Both are certified per block:
//   * |r_ref[j] - r[j]| <= delta = (n + 64) * 2^-53 * r[0]: the reference adds <= n/2 + 2 terms c*(a+b) one after the
//     other (each term two roundings, sum|terms| <= sum|x_i x_(i+lag)| <= r0), k_acf_blocks <= 48 (lane chain, wave tree);
//   * to first order the reflection coefficient of stage m moves by
//        dk_m = -(B' dR A)/e + k_m (A' dR A)/e,   A = (1, a_1 .. a_(m-1), 0), B = A reversed, e = e_(m-1)
//     (the optimal predictors make the variation of A and B drop out: R A and R B vanish in the middle rows), hence
//        |dk_m| <= ||a^(m-1)||_1^2 * (1 + |k_m|) / e_(m-1) * max_j |dr_j|;
//   * the rounding of the recursion itself (the reference's run and this one) enters stage m through its sum
//     num_m = sum a_i r_(m-i), <= (m+2) 2^-53 ||a||_1 r0 per run: it is put through the same sensitivity;
//   * `safety` (16) multiplies the whole bound: tests/tools/cert_study.py and tests/test_gpu_cert.py measure the
//     largest |k - k_ref| at 0.6 % of the UNscaled bound over tones with noise floors down to -140 dB, music-like and
//     full-scale material at orders 16/32/48.
// A block is certified when every k_m * 2^(q-1) keeps its distance from the rounding boundaries (half-integers, and
// the two clip limits) and the estimated code length keeps its distance from the RAW threshold; everything else --
// non-finite values, |k| >= 1, r0 near FLT_EPSILON -- is flagged: its group index goes to fb_list and the exact kernels
// (k_lpc_blocks + k_blocks_finish<.., false> in list mode) redo it on the same stream.
// list mode (list != NULL): the groups to finish are list[0 .. *list_count).
// ---------------------------------------------------------------------------------------------
template <int P, bool CERT>           // P >= order
__global__ __launch_bounds__(64)
void k_blocks_finish(const sla_hip_lpc_group* __restrict__ groups, uint32_t num_groups, uint32_t order,
                     double* __restrict__ out, int32_t* __restrict__ out_code, int32_t* __restrict__ out_kint,
                     const uint32_t* __restrict__ out_rshift,
                     const uint32_t* __restrict__ list, const uint32_t* __restrict__ list_count,
                     uint32_t* __restrict__ cert_flag, uint32_t* __restrict__ fb_list, uint32_t* __restrict__ fb_count,
                     double safety, uint32_t bps, const uint32_t* __restrict__ dyn = nullptr, uint32_t audit_every = 0u)
{
  // dyn: k_expand_scan's running numbers -- the launch was sized for the most groups the file can have, before the host
  // knew how many there are (see sla_hip_launch_lpc_blocks_cert)
  const uint32_t total = (list != nullptr) ? *list_count : ((dyn != nullptr) ? ((dyn[2] != 0u) ? 0u : (dyn[1] - dyn[3])) : num_groups);
  for (uint32_t li = blockIdx.x * 64 + threadIdx.x; li < total; li += gridDim.x * 64) {
  const uint32_t gi = (list != nullptr) ? list[li] : li;
  const uint32_t O1 = order + 1, O2 = order + 2;
  const sla_hip_lpc_group g = groups[gi];
  const uint64_t slot = g.slot_first;
  double* o = out + slot * O2;
  double r[P + 1], a[P + 1], par[P + 1];
#pragma unroll
  for (int i = 0; i <= P; i++) { r[i] = ((uint32_t)i <= order) ? o[1 + i] : 0.0; a[i] = 0.0; par[i] = 0.0; }
  const uint32_t rshift = out_rshift[slot];
  const double u = 1.1102230246251565e-16;      // 2^-53
  const double delta = CERT ? ((double)g.num_samples + 64.0) * u * r[0] : 0.0;
  bool sure = true;
  double gain = 0.0, gain_w = 0.0;              // sum log2(1 - k^2) and its half width
  // margin test of one coefficient: does Round(k * 2^(q-1)) (half-integers away from zero, then clipped) keep its value
  // for every k within +-eps?
#define SLA_CERT_COEF(m, kk, eps) do { \
    const double lim_ = ((m) < 4) ? 32768.0 : 128.0; \
    const double v_ = (kk) * lim_, w_ = (eps) * lim_ * 1.0000001 + 1e-12; \
    if (!(w_ < 0.25) || !(fabs(v_) < lim_ + 2.0)) { sure = false; } \
    else if (v_ >= lim_ - 1.5) { if (!(v_ - w_ > lim_ - 1.5)) { sure = false; } } \
    else if (v_ <= -lim_ + 0.5) { if (!(v_ + w_ < -lim_ + 0.5)) { sure = false; } } \
    else { \
      const double f_ = fabs(v_) + 0.5, d_ = f_ - floor(f_); \
      if (!(d_ > w_ && 1.0 - d_ > w_)) { sure = false; } \
      if (!(v_ + w_ < lim_ - 1.5) && !(v_ - w_ > lim_ - 1.5)) { sure = false; } \
      if (!(v_ - w_ > -lim_ + 0.5) && !(v_ + w_ < -lim_ + 0.5)) { sure = false; } \
    } \
    { const double om_ = 1.0 - (kk) * (kk); \
      if (!(om_ > 0.0)) { sure = false; } \
      else { gain += log2(om_); gain_w += 2.0 * fabs(kk) * (eps) / om_ * 1.4426950408889634; } } \
  } while (0)
  if (CERT && g.num_samples >= order && !(fabs(r[0]) - (double)FLT_EPSILON > 2.0 * delta) && !((double)FLT_EPSILON - fabs(r[0]) > 2.0 * delta)) { sure = false; }
  if (!(g.num_samples < order || fabs(r[0]) < (double)FLT_EPSILON)) {
    a[0] = 1.0;
    a[1] = -r[1] / r[0];
    par[1] = r[1] / r[0];
    double e = r[0] + r[1] * a[1];
    if (CERT) {
      const double eps = safety * (1.0 + fabs(par[1])) / r[0] * (delta + 6.0 * u * r[0]);
      SLA_CERT_COEF(1, par[1], eps);
    }
#pragma unroll
    for (int d = 1; d < P; d++) {
      if ((uint32_t)d < order) {
        double gamma = 0.0;
#pragma unroll
        for (int i = 0; i <= d; i++) { gamma += a[i] * r[d + 1 - i]; }
        double n1 = 0.0;
        if (CERT) {
#pragma unroll
          for (int i = 0; i <= d; i++) { n1 += fabs(a[i]); }
        }
        const double e_prev = e;
        gamma /= (-e);
        e = (1.0 - gamma * gamma) * e;
        // a[i] <- a[i] + gamma * v[i],  v = (0, a[d], a[d-1], .., a[1], 1),  a[0] = 1, a[d+1] = 0 beforehand
        double nw[P + 1];
#pragma unroll
        for (int i = 1; i <= d; i++) { nw[i] = a[i] + gamma * a[d + 1 - i]; }
#pragma unroll
        for (int i = 1; i <= d; i++) { a[i] = nw[i]; }
        a[0] = 1.0 + gamma * 0.0;
        a[d + 1] = 0.0 + gamma * 1.0;
        par[d + 1] = -gamma;
        if (CERT) {
          if (!(e_prev > 0.0) || !(fabs(gamma) < 1.0)) { sure = false; }
          const double lev = 2.0 * (double)(d + 3) * u * n1 * r[0];
          const double eps = safety * n1 * n1 * (1.0 + fabs(gamma)) / e_prev * (delta + lev);
          SLA_CERT_COEF(d + 1, -gamma, eps);
        }
      }
    }
  }
#undef SLA_CERT_COEF
  o[0] = r[0];
  out_code[slot * O1] = 0; out_kint[slot * O1] = 0;
  bool codes_same = true;                       // (list mode on an audited pair: what the certified run stored against this run)
#pragma unroll
  for (int j = 0; j <= P; j++) {
    if ((uint32_t)j <= order) {
      o[1 + j] = par[j];
      if (j >= 1) {
        const uint32_t qb = (j < 4) ? 16 : 8;
        const int32_t lim = 1 << (qb - 1);
        const double kq = par[j] * (double)lim;
        const double rk = (kq >= 0.0) ? floor(kq + 0.5) : -floor(-kq + 0.5);
        int32_t code = f64_to_i32_x86(rk);
        code = (code < -lim) ? -lim : code;
        code = (code > lim - 1) ? (lim - 1) : code;
        const int32_t ki = (int32_t)((uint32_t)code << (16u - qb)) >> rshift;
        if (!CERT && cert_flag != nullptr && (out_code[slot * O1 + j] != code || out_kint[slot * O1 + j] != ki)) { codes_same = false; }
        out_code[slot * O1 + j] = code;
        out_kint[slot * O1 + j] = ki;
      }
    }
  }
  bool raw_side = false;
  if (CERT) {
    // RAW decision of the host (slai_code_length, src/SLAPredictor.c:416-468; threshold (double)0.95f on 8*bytes/bps):
    // bits = 1.94.. + (log2(r0 * 2^(2(bps-1)) / n) + sum log2(1 - k^2)) / 2, known here to +- half the widths
    const double power = r[0] * ldexp(1.0, (int)(2 * (bps - 1)));
    if (power > (double)FLT_MIN * 1.000001) {
      const double bits = 1.9426950408889634 + 0.5 * (log2(power) - log2((double)g.num_samples) + gain);
      const double half = 0.5 * (delta / r[0] * 1.4426950408889634 * 2.0 + gain_w) * 1.000001 + 1e-9;
      const double thr = (double)0.95f * (double)bps;
      if (!(fabs(bits - thr) > half)) { sure = false; }
      raw_side = (bits >= thr);
    } else if (!(power < (double)FLT_MIN * 0.999999)) { sure = false; }
    if (!(r[0] == r[0]) || !(fabs(r[0]) < 1e300)) { sure = false; }
    // audit (option cert_audit = N): every N-th CERTIFIED pair goes on the list as well, after its codes are stored; the exact
    // kernels then compare instead of overwrite (flag 3 | 8 * "the estimated length is on the RAW side of the threshold")
    const bool audit = sure && audit_every != 0u && (gi % audit_every) == audit_every - 1u;
    cert_flag[slot] = !sure ? 1u : !audit ? 0u : (3u | (raw_side ? 8u : 0u));
    if (!sure || audit) { fb_list[atomicAdd(fb_count, 1u)] = gi; }
  } else if (cert_flag != nullptr) {
    // exact: the reference's doubles bit for bit (2).  An audited pair (3): codes, lattice coefficients and the side of the RAW
    // threshold the certified run stored must be what this run computes -- 4 = they are, 5 = the certificate was WRONG
    // (the host fails the call).  The comparison of the estimated length ignores differences below 1e-6 bit (device log2
    // against itself on two sets of doubles that agree to ~1e-13: only a decision the certificate should never have taken
    // can differ by more).
    const uint32_t was = cert_flag[slot];
    if ((was & 7u) == 3u) {
      bool same = codes_same;
      if (bps != 0u) {
        double gain_e = 0.0;
        bool pd = true;
#pragma unroll
        for (int j = 1; j <= P; j++) { if ((uint32_t)j <= order) { const double om = 1.0 - par[j] * par[j]; if (om > 0.0) { gain_e += log2(om); } else { pd = false; } } }
        const double power = r[0] * ldexp(1.0, (int)(2 * (bps - 1)));
        if (pd && power > (double)FLT_MIN * 1.000001) {
          const double bits = 1.9426950408889634 + 0.5 * (log2(power) - log2((double)g.num_samples) + gain_e);
          const double thr = (double)0.95f * (double)bps;
          if (fabs(bits - thr) > 1e-6 && (bits >= thr) != ((was & 8u) != 0u)) { same = false; }
        } else if (!pd) { same = false; }
      }
      cert_flag[slot] = same ? 4u : 5u;
    } else {
      cert_flag[slot] = 2u;
    }
  }
  }
}

// ---------------------------------------------------------------------------------------------
// Exact partition search (see include/sla_hip.h): k_acf_tiles + k_search_finish.
//
// k_acf_tiles: one wave per (group, 1024-sample tile).  A lane owns 4 consecutive samples m[0..3]; the
// partner values x[m+lag] of lag block 4k..4k+3 are the 4-sample runs of lanes t+k and t+k+1, which
// arrive by a one-lane DPP shift per block -- 16 FMAs per 8 shifted dwords.  The last NB lanes
// of a pass only supply partners (their own multipliers are zero), so a pass advances (64-NB)*4
// samples; all passes' samples are requested before the first one is used.  Partners beyond the
// group's window are zero, partners beyond the tile are not: P[lag] holds every pair whose FIRST sample
// lies in the tile, X[lag] the ones among them whose second sample lies beyond it (a candidate ending
// with this tile has to give them back).
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ double shl1_f64(double v)       // lane t <- lane t+1, lane 63 <- 0
{
  const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), 0x130 /* wave_shl:1 */, 0xF, 0xF, true);
  const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), 0x130, 0xF, 0xF, true);
  return __hiloint2double(hi, lo);
}

template <int CTRL>
__device__ __forceinline__ double dpp_f64(double v)        // all source lanes valid for the row permutations used
{
  const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), CTRL, 0xF, 0xF, true);
  const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), CTRL, 0xF, 0xF, true);
  return __hiloint2double(hi, lo);
}

// sum over the 64 lanes, valid in every lane (order of the additions is free: callers' sums are exact).
// In-row steps are DPP moves; the four row totals travel through SGPRs (ds_bpermute measured ~8x dearer).
__device__ __forceinline__ double wave_sum_f64(double v)
{
  v += dpp_f64<0xB1>(v);        // quad_perm [1,0,3,2]
  v += dpp_f64<0x4E>(v);        // quad_perm [2,3,0,1]
  v += dpp_f64<0x141>(v);       // row_half_mirror
  v += dpp_f64<0x140>(v);       // row_mirror
  const int lo = __double2loint(v), hi = __double2hiint(v);
  const double r0 = __hiloint2double(__builtin_amdgcn_readlane(hi, 0), __builtin_amdgcn_readlane(lo, 0));
  const double r1 = __hiloint2double(__builtin_amdgcn_readlane(hi, 16), __builtin_amdgcn_readlane(lo, 16));
  const double r2 = __hiloint2double(__builtin_amdgcn_readlane(hi, 32), __builtin_amdgcn_readlane(lo, 32));
  const double r3 = __hiloint2double(__builtin_amdgcn_readlane(hi, 48), __builtin_amdgcn_readlane(lo, 48));
  return (r0 + r1) + (r2 + r3);
}

// Sums over the 64 lanes of M values per lane at once (M = 16, 32, 64; callers pad with zeros): a butterfly that halves
// the number of live values at every step -- a lane keeps the half selected by one of its lane-number bits and hands
// the other half to the partner that differs in that bit -- so the whole job costs M - 1 exchange-and-add operations
// instead of M separate 6-step wave sums.  The first (widest) steps use the cheap in-row DPP exchanges (xor 1, xor 2,
// xor 8 = rotate by 8 inside a row of 16), the narrow ones ds_swizzle / bpermute.  On return lane L holds the total of
// value `index` (every index < M exactly once among the lanes `writer` marks).  Callers' sums are exact or certified
// for any order of the additions.
__device__ __forceinline__ double swz_f64_xor4(double v)
{
  return __hiloint2double(__builtin_amdgcn_ds_swizzle(__double2hiint(v), 0x101F), __builtin_amdgcn_ds_swizzle(__double2loint(v), 0x101F));
}
__device__ __forceinline__ double swz_f64_xor16(double v)
{
  return __hiloint2double(__builtin_amdgcn_ds_swizzle(__double2hiint(v), 0x401F), __builtin_amdgcn_ds_swizzle(__double2loint(v), 0x401F));
}
__device__ __forceinline__ double shfl_f64_xor32(double v)
{
  return __hiloint2double(__shfl_xor(__double2hiint(v), 32), __shfl_xor(__double2loint(v), 32));
}
template <int STEP>
__device__ __forceinline__ double lane_exchange(double v)       // partner = lane ^ {1, 2, 8, 4, 16, 32}[STEP]
{
  if (STEP == 0) { return dpp_f64<0xB1>(v); }
  if (STEP == 1) { return dpp_f64<0x4E>(v); }
  if (STEP == 2) { return dpp_f64<0x128>(v); }                  // row_ror:8
  if (STEP == 3) { return swz_f64_xor4(v); }
  if (STEP == 4) { return swz_f64_xor16(v); }
  return shfl_f64_xor32(v);
}
template <int M, int STEP>
__device__ __forceinline__ void transpose_step(double (&v)[M], uint32_t lane)
{
  constexpr int K = M >> STEP;                                  // live values before this step
  constexpr uint32_t BIT = (STEP == 0) ? 1u : (STEP == 1) ? 2u : (STEP == 2) ? 8u : (STEP == 3) ? 4u : (STEP == 4) ? 16u : 32u;
  const bool up = (lane & BIT) != 0;
  if (K >= 2) {
#pragma unroll
    for (int j = 0; j < K / 2; j++) {
      const double lo = v[j], hi = v[j + K / 2];
      const double send = up ? lo : hi, keep = up ? hi : lo;
      v[j] = keep + lane_exchange<STEP>(send);
    }
  } else {
    v[0] += lane_exchange<STEP>(v[0]);                          // one value left: plain sum over the remaining lane bits
  }
}
template <int M>
__device__ __forceinline__ double wave_transpose_sum(double (&v)[M], uint32_t lane, uint32_t& index, bool& writer)
{
  transpose_step<M, 0>(v, lane); transpose_step<M, 1>(v, lane); transpose_step<M, 2>(v, lane);
  transpose_step<M, 3>(v, lane); transpose_step<M, 4>(v, lane); transpose_step<M, 5>(v, lane);
  const uint32_t b0 = lane & 1u, b1 = (lane >> 1) & 1u, b2 = (lane >> 2) & 1u, b3 = (lane >> 3) & 1u, b4 = (lane >> 4) & 1u, b5 = (lane >> 5) & 1u;
  // step k keeps the half selected by its bit: the bits spell the index from the top (M/2, M/4, ..)
  if (M == 64) { index = b0 * 32 + b1 * 16 + b3 * 8 + b2 * 4 + b4 * 2 + b5; writer = true; }
  else if (M == 32) { index = b0 * 16 + b1 * 8 + b3 * 4 + b2 * 2 + b4; writer = (b5 == 0); }
  else { index = b0 * 8 + b1 * 4 + b3 * 2 + b2; writer = (b4 == 0 && b5 == 0); }
  return v[0];
}

typedef int32_t i32x4_u __attribute__((ext_vector_type(4), aligned(4)));

// 4 consecutive raw samples of one plane, zero from `limit` on
__device__ __forceinline__ void load4_raw(const int32_t* __restrict__ plane, uint64_t base, uint32_t idx, uint32_t limit, int32_t (&v)[4])
{
  if (idx + 3 < limit) {
    const i32x4_u t = *(const i32x4_u*)(plane + base + idx);
    v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w;
  } else {
#pragma unroll
    for (int q = 0; q < 4; q++) { v[q] = (idx + q < limit) ? plane[base + idx + q] : 0; }
  }
}

// ---------------------------------------------------------------------------------------------
// The sliding dot products of a 256-sample tile, partners from LDS: lane t owns x[4t .. 4t+3] and pairs every one of them
// with the `lag` samples BEHIND it -- r[lag] = sum x[m] x[m - lag] -- so a tile only needs the samples in front of it, which
// the tile before has already staged, and the loads of the next tile can travel while this one is multiplied.  For the lag
// block 4k .. 4k+3 a lane needs x[4(t-k) - 3 .. 4(t-k) + 3].  The samples lie in LDS as 16-byte pairs, even pairs
// (x[4i], x[4i+1]) in E and odd pairs (x[4i+2], x[4i+3]) in O, slot HL + i for the tile's own pair i and slots [0, HL) for
// the HL = NB + 1 pairs in front of the tile: consecutive lanes read consecutive 16-byte slots (no bank conflicts), two
// ds_read_b128 per 16 FMAs.  (The first form of these kernels moved the partners from lane to lane by DPP: 16 moves per
// 16 FMAs, the last NB lanes of a pass only supplied partners, and nothing was in flight while a pass computed.)
// ---------------------------------------------------------------------------------------------
template <int NB>
__device__ __forceinline__ void acf_tile_fma(const double (&own)[4], const double2* __restrict__ E, const double2* __restrict__ O,
                                             uint32_t lane, double (&acc)[NB * 4])
{
  constexpr uint32_t HL = NB + 1;
  const double2* e = E + HL + lane;
  const double2* o = O + HL + lane;
  // block k works on P[i] = x[4(t-k) - 3 + i], i = 0 .. 6: (E[-k-1].y, O[-k-1].x, O[-k-1].y, E[-k].x, E[-k].y, O[-k].x, O[-k].y)
  double2 e0 = make_double2(own[0], own[1]), o0 = make_double2(own[2], own[3]);
  double2 e1 = e[-1], o1 = o[-1];
#pragma unroll
  for (int k = 0; k < NB; k++) {
    double2 e2 = e1, o2 = o1;
    if (k + 1 < NB) { e2 = e[-k - 2]; o2 = o[-k - 2]; }                    // the next block's new pairs travel under this block's FMAs
    const double P[7] = {e1.y, o1.x, o1.y, e0.x, e0.y, o0.x, o0.y};
#pragma unroll
    for (int j = 0; j < 4; j++) {
#pragma unroll
      for (int q = 0; q < 4; q++) { acc[4 * k + j] = __builtin_fma(own[q], P[q - j + 3], acc[4 * k + j]); }
    }
    e0 = e1; o0 = o1; e1 = e2; o1 = o2;
  }
}

#define ACF_TILE 256u

// Small device words that have to be zero before the kernels BEHIND this one on the stream use them (counters, flags, the
// execution spans): cleared by the first workgroup of a kernel that is launched anyway instead of by a memset each -- a
// 4-byte hipMemsetAsync is a 4 us fill kernel plus a launch boundary, and a ten-minute mono file had five of them on its
// critical path (45 us of a 1.2 ms step).
struct clear_list { uint32_t* ptr[4]; uint32_t words[4]; };
__device__ __forceinline__ void clear_words(const clear_list& cl)
{
  if (blockIdx.x != 0) { return; }
#pragma unroll
  for (int r = 0; r < 4; r++) {
    if (cl.ptr[r] != nullptr) { for (uint32_t i = threadIdx.x; i < cl.words[r]; i += blockDim.x) { cl.ptr[r][i] = 0u; } }
  }
}

template <int NB>
__global__ __launch_bounds__(256)
void k_acf_tiles(const int32_t* __restrict__ pcm, uint64_t stride, uint32_t ms,
                 const sla_hip_lpc_group* __restrict__ groups, uint32_t num_groups, uint32_t tiles_per_group,
                 double* __restrict__ tile_sums, clear_list cl)
{
  constexpr uint32_t OL = 64 - NB;             // lanes of a pass that own samples
  constexpr uint32_t STEP = OL * 4, PASSES = (SLA_HIP_XTILE + STEP - 1) / STEP, LAGS = NB * 4;
  __shared__ double s_edge[4][2 * LAGS];       // x[t1-LAGS .. t1+LAGS) of each wave's tile end t1
  clear_words(cl);
  const uint32_t lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const uint32_t w = blockIdx.x * 4 + wv;
  const uint32_t gi = w / tiles_per_group, tile = w - gi * tiles_per_group;
  if (gi >= num_groups) { return; }
  const sla_hip_lpc_group g = groups[gi];
  const uint32_t N = g.num_samples, t0 = tile * SLA_HIP_XTILE;
  if (t0 >= N) { return; }
  const uint32_t t1 = (t0 + SLA_HIP_XTILE < N) ? (t0 + SLA_HIP_XTILE) : N;
  const double scale = 4.656612873077392578125e-10;   // 2^-31, exact (same arithmetic as load_f64)

  int32_t ra[PASSES][4], rb[PASSES][4];
#pragma unroll
  for (uint32_t p = 0; p < PASSES; p++) {
    const uint32_t idx = t0 + p * STEP + 4 * lane;
    if (t0 + p * STEP < t1) {
      load4_raw(pcm + (ms ? 0 : (uint64_t)g.channel * stride), g.pcm_off, idx, N, ra[p]);
      if (ms) { load4_raw(pcm + stride, g.pcm_off, idx, N, rb[p]); }
    }
  }
  for (uint32_t i = lane; i < 2 * LAGS; i += 64) { s_edge[wv][i] = 0.0; }

  double acc[LAGS];
#pragma unroll
  for (int i = 0; i < (int)LAGS; i++) { acc[i] = 0.0; }
#pragma unroll
  for (uint32_t p = 0; p < PASSES; p++) {
    const uint32_t s0 = t0 + p * STEP;
    if (s0 < t1) {
      double cur[4], own[4];
#pragma unroll
      for (int q = 0; q < 4; q++) {
        const uint32_t idx = s0 + 4 * lane + q;
        if (ms) {
          const double l = (double)ra[p][q] * scale, r = (double)rb[p][q] * scale;
          cur[q] = (g.channel == 0) ? ((l + r) / 2) : (l - r);
        } else {
          cur[q] = (double)ra[p][q] * scale;
        }
        own[q] = (lane < OL && idx < t1) ? cur[q] : 0.0;
        const int rel = (int)idx - ((int)t1 - (int)LAGS);
        if (rel >= 0 && rel < (int)(2 * LAGS)) { s_edge[wv][rel] = cur[q]; }
      }
#pragma unroll
      for (int k = 0; k < NB; k++) {
        double nxt[4];
#pragma unroll
        for (int q = 0; q < 4; q++) { nxt[q] = shl1_f64(cur[q]); }
#pragma unroll
        for (int j = 0; j < 4; j++) {
#pragma unroll
          for (int q = 0; q < 4; q++) {
            const double partner = (q + j < 4) ? cur[q + j] : nxt[q + j - 4];
            acc[4 * k + j] = __builtin_fma(own[q], partner, acc[4 * k + j]);     // exact below the limit: fusing is free
          }
        }
#pragma unroll
        for (int q = 0; q < 4; q++) { cur[q] = nxt[q]; }
      }
    }
  }
  double* dst = tile_sums + ((uint64_t)gi * SLA_HIP_XTILES + tile) * (2 * LAGS);
  {
    constexpr int M = (LAGS <= 16) ? 16 : (LAGS <= 32) ? 32 : 64;
    double tv[M];
#pragma unroll
    for (int i = 0; i < M; i++) { tv[i] = (i < (int)LAGS) ? acc[i] : 0.0; }
    uint32_t index; bool writer;
    const double total = wave_transpose_sum<M>(tv, lane, index, writer);
    if (writer && index < LAGS) { dst[index] = total; }
  }
  // pairs that straddle t1: lane = lag
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  if (lane < LAGS) {
    double x = 0.0;
    const double* e = s_edge[wv];
    for (uint32_t j = 0; j < lane; j++) { x = __builtin_fma(e[LAGS - lane + j], e[LAGS + j], x); }
    dst[LAGS + lane] = x;
  }
}

// ---------------------------------------------------------------------------------------------
// k_acf_tiles_lds: the tile sums of k_acf_tiles with the partners from LDS (acf_tile_fma) instead of DPP moves -- for 52
// lags, where the moves are most of the instructions (C5: 489 -> see DESIGN us per launch; at 36 lags and below the DPP
// kernel is as fast or faster and stays).  acf_tile_fma pairs a sample with the ones BEHIND it; P_t[lag] needs the ones
// AHEAD (x[m] x[m + lag], m in the tile, m + lag anywhere in the window), so the tile is walked BACKWARDS: in reversed
// coordinates u = (t1 + 4 HL - 1) - m the partner m + lag is u - lag, the first 4 HL reversed samples -- the ones just
// above the tile's end -- are the front nobody owns, and the tile's own samples follow as up to four sub-tiles of 256.
// Samples below t0 count as zero (they are another tile's), samples from the window's end on are zero anyway.  X_t as
// in k_acf_tiles, from an edge buffer.  (The first LDS port of round 3 kept the forward walk and re-defined P and X by
// the position of a pair's LATER sample; it was three times slower and was dropped.)
// ---------------------------------------------------------------------------------------------
template <int NB>
__global__ __launch_bounds__(256, (NB >= 13) ? 3 : 1)
void k_acf_tiles_lds(const int32_t* __restrict__ pcm, uint64_t stride, uint32_t ms,
                     const sla_hip_lpc_group* __restrict__ groups, uint32_t num_groups, uint32_t tiles_per_group,
                     double* __restrict__ tile_sums, clear_list cl)
{
  constexpr uint32_t LAGS = NB * 4, HL = NB + 1;
  __shared__ double2 s_e[4][2][64 + HL], s_o[4][2][64 + HL];
  __shared__ double s_edge[4][2 * LAGS];       // x[t1-LAGS .. t1+LAGS) of each wave's tile end t1 (zero below t0)
  clear_words(cl);
  const uint32_t lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const uint32_t w = blockIdx.x * 4 + wv;
  const uint32_t gi = w / tiles_per_group, tile = w - gi * tiles_per_group;
  if (gi >= num_groups) { return; }
  const sla_hip_lpc_group g = groups[gi];
  const uint32_t N = g.num_samples, t0 = tile * SLA_HIP_XTILE;
  if (t0 >= N) { return; }
  const uint32_t t1 = (t0 + SLA_HIP_XTILE < N) ? (t0 + SLA_HIP_XTILE) : N;
  const double scale = 4.656612873077392578125e-10;   // 2^-31, exact (same arithmetic as load_f64)
  const int32_t* p0 = pcm + (ms ? 0 : (uint64_t)g.channel * stride) + g.pcm_off;
  const int32_t* p1 = pcm + stride + g.pcm_off;
  auto value = [&](int32_t a, int32_t b) -> double {
    if (ms) { const double l = (double)a * scale, r = (double)b * scale; return (g.channel == 0) ? ((l + r) / 2) : (l - r); }
    return (double)a * scale;
  };
  // four consecutive samples m0 .. m0+3 (m0 may be negative), zero outside [lo, N)
  struct raw4 { int32_t a[4]; int32_t b[4]; };
  auto fetch = [&](int64_t m0, uint32_t lo, raw4& r) {
#pragma unroll
    for (int q = 0; q < 4; q++) { r.a[q] = 0; r.b[q] = 0; }
    if (m0 >= (int64_t)lo && m0 + 3 < (int64_t)N) {
      const i32x4_u t = *(const i32x4_u*)(p0 + m0);
      r.a[0] = t.x; r.a[1] = t.y; r.a[2] = t.z; r.a[3] = t.w;
      if (ms) { const i32x4_u u = *(const i32x4_u*)(p1 + m0); r.b[0] = u.x; r.b[1] = u.y; r.b[2] = u.z; r.b[3] = u.w; }
    } else {
#pragma unroll
      for (int q = 0; q < 4; q++) {
        const int64_t m = m0 + q;
        if (m >= (int64_t)lo && m < (int64_t)N) { r.a[q] = p0[m]; if (ms) { r.b[q] = p1[m]; } }
      }
    }
  };
  // edge buffer for X
  for (uint32_t i = lane; i < 2 * LAGS; i += 64) {
    const int64_t m = (int64_t)t1 - (int64_t)LAGS + (int64_t)i;
    double v = 0.0;
    if (m >= (int64_t)t0 && m < (int64_t)N) { v = value(p0[m], ms ? p1[m] : 0); }
    s_edge[wv][i] = v;
  }
  // front: the 4 HL samples from t1 on, reversed; lane i < HL holds reversed positions 4 i .. 4 i + 3 = x[t1 + 4 HL - 1 - 4 i] downwards
  if (lane < HL) {
    raw4 f;
    fetch((int64_t)t1 + 4 * (int64_t)HL - 4 - 4 * (int64_t)lane, t0, f);
    s_e[wv][0][lane] = make_double2(value(f.a[3], f.b[3]), value(f.a[2], f.b[2]));
    s_o[wv][0][lane] = make_double2(value(f.a[1], f.b[1]), value(f.a[0], f.b[0]));
  }
  double acc[LAGS];
#pragma unroll
  for (int i = 0; i < (int)LAGS; i++) { acc[i] = 0.0; }
  const uint32_t nsub = (t1 - t0 + ACF_TILE - 1) / ACF_TILE;
  raw4 nxt;
  fetch((int64_t)t1 - (int64_t)ACF_TILE + 4 * (int64_t)(63 - lane), t0, nxt);
  uint32_t buf = 0;
  for (uint32_t sub = 0; sub < nsub; sub++, buf ^= 1u) {
    const raw4 cur = nxt;
    if (sub + 1 < nsub) { fetch((int64_t)t1 - (int64_t)ACF_TILE * (sub + 2) + 4 * (int64_t)(63 - lane), t0, nxt); }
    double own[4];
#pragma unroll
    for (int q = 0; q < 4; q++) { own[q] = value(cur.a[3 - q], cur.b[3 - q]); }      // reversed: the lane's highest sample first
    double2* E = s_e[wv][buf];
    double2* O = s_o[wv][buf];
    E[HL + lane] = make_double2(own[0], own[1]);
    O[HL + lane] = make_double2(own[2], own[3]);
    if (lane >= 64 - HL) {                     // this sub-tile's last HL pairs are what the next one finds in front of it
      s_e[wv][buf ^ 1u][lane - (64 - HL)] = make_double2(own[0], own[1]);
      s_o[wv][buf ^ 1u][lane - (64 - HL)] = make_double2(own[2], own[3]);
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    acf_tile_fma<NB>(own, E, O, lane, acc);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  }
  double* dst = tile_sums + ((uint64_t)gi * SLA_HIP_XTILES + tile) * (2 * LAGS);
  {
    constexpr int M = (LAGS <= 16) ? 16 : (LAGS <= 32) ? 32 : 64;
    double tv[M];
#pragma unroll
    for (int i = 0; i < M; i++) { tv[i] = (i < (int)LAGS) ? acc[i] : 0.0; }
    uint32_t index; bool writer;
    const double total = wave_transpose_sum<M>(tv, lane, index, writer);
    if (writer && index < LAGS) { dst[index] = total; }
  }
  // pairs that straddle t1: lane = lag
  if (lane < LAGS) {
    double x = 0.0;
    const double* e = s_edge[wv];
    for (uint32_t j = 0; j < lane; j++) { x = __builtin_fma(e[LAGS - lane + j], e[LAGS + j], x); }
    dst[LAGS + lane] = x;
  }
}

// ---------------------------------------------------------------------------------------------
// k_acf_blocks: autocorrelation of the CHOSEN blocks' analysis windows in any summation order (the certified route of
// the block stage, see k_blocks_finish<.., true>).  One wave per (block, channel) walks the whole window in tiles of 256
// samples (acf_tile_fma); the accumulators live in registers across the tiles and are reduced over the lanes ONCE per
// block.  The samples are staged as the reference stages them -- x[s] = w[s]*in[s] - 0.96875 * w[s-1]*in[s-1]
// (src/SLAEncoder.c:505-515, 540-543, src/SLAPredictor.c:1803-1809): the previous windowed sample comes from the lane
// below (from the last lane of the tile before for lane 0).  Out: slot = { -, r[0..order] } and the quantiser's shift,
// exactly what k_lpc_blocks hands to k_blocks_finish -- but r[] is the correctly ordered sum only up to the rounding
// errors of ~48 additions.
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ double shr1_f64(double v, double first)       // lane t <- lane t-1, lane 0 <- first
{
  const int lo = __builtin_amdgcn_update_dpp(__double2loint(first), __double2loint(v), 0x138 /* wave_shr:1 */, 0xF, 0xF, false);
  const int hi = __builtin_amdgcn_update_dpp(__double2hiint(first), __double2hiint(v), 0x138, 0xF, 0xF, false);
  return __hiloint2double(hi, lo);
}

template <int NB, bool MS>
__global__ __launch_bounds__(256, (NB >= 13) ? 3 : 1)      // 52 lags: 104 accumulator registers -- keep three waves per SIMD
void k_acf_blocks(const int32_t* __restrict__ pcm, uint64_t stride, uint32_t order,
                  const sla_hip_lpc_group* __restrict__ groups, uint32_t num_groups,
                  const double* __restrict__ window_pool, double* __restrict__ out, uint32_t* __restrict__ out_rshift,
                  unsigned long long* exec_span, uint32_t* __restrict__ zero_word, const uint32_t* __restrict__ dyn)
{
  constexpr uint32_t LAGS = NB * 4, HL = NB + 1;             // HL pairs in front of a tile: all that lags < 4 NB reach
  __shared__ double2 s_e[4][2][64 + HL], s_o[4][2][64 + HL];
  if (dyn != nullptr) { num_groups = (dyn[2] != 0u) ? 0u : (dyn[1] - dyn[3]); }      // (see k_blocks_finish)
  span_begin(exec_span);
  if (blockIdx.x == 0 && threadIdx.x == 0 && zero_word != nullptr) { *zero_word = 0u; }      // the fallback count k_blocks_finish appends to
  const uint32_t lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const uint32_t gi = blockIdx.x * 4 + wv;
  if (gi >= num_groups) { span_end(exec_span); return; }
  const sla_hip_lpc_group g = groups[gi];
  const uint32_t N = g.num_samples;
  const double scale = 4.656612873077392578125e-10;   // 2^-31, exact
  const int32_t* p0 = pcm + (MS ? 0 : (uint64_t)g.channel * stride);
  const int32_t* p1 = pcm + stride;
  const double* win = window_pool + g.win_off;

  struct raw4 { int32_t a[4]; int32_t b[4]; double w[4]; };
  auto fetch = [&](uint32_t idx, raw4& r) {      // four consecutive raw samples (both channels for mid/side) and their window values; zero past the block
    load4_raw(p0, g.pcm_off, idx, N, r.a);
    if (MS) { load4_raw(p1, g.pcm_off, idx, N, r.b); }
#pragma unroll
    for (int q = 0; q < 4; q++) { r.w[q] = (idx + q < N) ? win[idx + q] : 0.0; }
  };

  double acc[LAGS];
#pragma unroll
  for (int i = 0; i < (int)LAGS; i++) { acc[i] = 0.0; }
  uint32_t maxabs = 0;
  double carry = 0.0;                          // windowed sample right before the tile (0 before the block: src/SLAPredictor.c:1729-1738)
  if (lane < HL) { s_e[wv][0][lane] = make_double2(0.0, 0.0); s_o[wv][0][lane] = make_double2(0.0, 0.0); }      // nothing in front of the block
  raw4 nxt;
  fetch(4 * lane, nxt);
  uint32_t buf = 0;
  for (uint32_t s0 = 0; s0 < N; s0 += ACF_TILE, buf ^= 1u) {
    const uint32_t idx = s0 + 4 * lane;
    const raw4 cur = nxt;
    if (s0 + ACF_TILE < N) { fetch(idx + ACF_TILE, nxt); }      // the next tile's samples travel while this one is multiplied
    double y[4], own[4];
#pragma unroll
    for (int q = 0; q < 4; q++) {
      double v;
      int32_t iv;
      if (MS) {
        const double l = (double)cur.a[q] * scale, r = (double)cur.b[q] * scale;
        v = (g.channel == 0) ? ((l + r) / 2) : (l - r);
        const int32_t li = cur.a[q] >> g.int_shift, ri = cur.b[q] >> g.int_shift;
        iv = (g.channel == 0) ? ((int32_t)((uint32_t)li + (uint32_t)ri) >> 1) : (int32_t)((uint32_t)li - (uint32_t)ri);
      } else {
        v = (double)cur.a[q] * scale;
        iv = cur.a[q] >> g.int_shift;
      }
      y[q] = v * cur.w[q];
      const uint32_t a = (iv > 0) ? (uint32_t)iv : (0u - (uint32_t)iv);
      maxabs = (a > maxabs) ? a : maxabs;        // (samples past the block are zero)
    }
    const double below = shr1_f64(y[3], carry);
#pragma unroll
    for (int q = 0; q < 4; q++) {
      const double pv = (q == 0) ? below : y[q - 1];
      const double x = y[q] - pv * 0.96875;
      own[q] = (idx + q < N) ? x : 0.0;
    }
    carry = readlane_f64(y[3], 63);
    double2* E = s_e[wv][buf];
    double2* O = s_o[wv][buf];
    E[HL + lane] = make_double2(own[0], own[1]);
    O[HL + lane] = make_double2(own[2], own[3]);
    if (lane >= 64 - HL) {                     // this tile's last HL pairs are what the next tile finds in front of it
      s_e[wv][buf ^ 1u][lane - (64 - HL)] = make_double2(own[0], own[1]);
      s_o[wv][buf ^ 1u][lane - (64 - HL)] = make_double2(own[2], own[3]);
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    acf_tile_fma<NB>(own, E, O, lane, acc);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  }
  double* o = out + (uint64_t)g.slot_first * (order + 2);
  {
    constexpr int M = (LAGS <= 16) ? 16 : (LAGS <= 32) ? 32 : 64;
    double tv[M];
#pragma unroll
    for (int i = 0; i < M; i++) { tv[i] = (i < (int)LAGS) ? acc[i] : 0.0; }
    uint32_t index; bool writer;
    const double total = wave_transpose_sum<M>(tv, lane, index, writer);
    if (writer && index <= order) { o[1 + index] = total; }
  }
  maxabs = umax_wave(maxabs);
  if (lane == 0) {
    const uint32_t l2c = (maxabs > 1) ? (32u - (uint32_t)__builtin_clz(maxabs - 1u)) : 0u;    // src/SLAUtility.c:677-696
    const uint32_t bitwidth = (maxabs > 0) ? (l2c + 1u) : 1u;
    out_rshift[g.slot_first] = (bitwidth > 16) ? (bitwidth - 16) : 0;
  }
  span_end(exec_span);
}

// k_search_finish: one wave per group.  r[lag] of candidate [start, end) = P of its tiles minus X of its
// last tile, then Levinson-Durbin per candidate exactly as in k_lpc.
//
// Windows over the exactness limit (loud material wider than 16 bits): the tile sums no longer equal the reference's
// serially rounded sums bit for bit -- but the search only has to deliver a PARTITION, and the reference's decision
// can be certified from sums that are merely close (cert > 0):
//   * the reference's r[lag] differs from the exact sum by at most n*2^-53 * sum|terms| <= n*2^-53 * r0 (recursive
//     summation of n exactly representable terms), ours by at most 48*2^-53 * (energy of the window) (<= 48 roundings
//     per lag: lane chain, wave tree, tiles), so the two Toeplitz matrices differ by a symmetric Toeplitz matrix whose
//     2-norm is at most (2*order+1) * delta;
//   * the estimated code length of a candidate depends on its autocorrelation only through the final prediction
//     error e_p = r0 * prod(1 - k_j^2) = min over a, a_0 = 1, of a'Ra, which is monotone in the Loewner order:
//     R~ - d*I <= R_ref <= R~ + d*I  implies  e_p(R~ - d*I) <= e_p(R_ref) <= e_p(R~ + d*I);
//   * d = cert * (2*order+1) * delta with cert = 64: a factor 63 on top of the summation bounds for the rounding of the
//     Levinson recursion itself on either side (measured: the reference's estimate sits within 1e-4 of the bracket's
//     half width, tests/test_gpu_parity.py::test_search_certificate_brackets_the_reference).
// The candidate's slot then carries the tile-sum result plus, in parcor[0] (always 0 otherwise), the half width of
// log2(e_p); k_plan adds the widths up along the paths and only accepts a partition no width can change.  Anything
// else (not positive definite, not finite, cert <= 0) is flagged -- NaN in r[0], or an infinite width -- and redone
// as serial chains.
#define XF_BATCH 64          // candidates one pass of the wave takes (at most)
#define XF_GROUPS 16         // groups one wave takes (at most)

// Final prediction error e_p of the Toeplitz system (r0, rc[1..order]) by the Schur recursion, both generator vectors
// in registers (static indices: the lower one moves down one slot per stage instead of the upper one moving up).
// Only used where nothing has to match the reference's Levinson recursion bit for bit (the certificate): fused
// multiply-adds, and NaN as soon as the recursion leaves the positive-definite range.
template <int P>
__device__ __forceinline__ double schur_error(const double* __restrict__ rc, double r0, uint32_t order, double& growth)
{
  const double nan = __longlong_as_double(0x7FF8000000000000ll);
  double u[P + 1], v[P + 2];
#pragma unroll
  for (int i = 0; i <= P; i++) { const double x = ((uint32_t)i <= order && i >= 1) ? rc[i] : 0.0; u[i] = (i == 0) ? r0 : x; v[i] = x; }
  v[P + 1] = 0.0;
  bool bad = false;
  // stage m only has to renew the entries the later stages still read: u[0 .. P-m], v[0 .. P-m] (the result's cone of
  // dependence shrinks by one index per stage) -- half the multiply-adds of a square sweep; stages and indices are
  // unrolled, so both generator vectors stay in registers
#pragma unroll
  for (int m = 1; m <= P; m++) {
    if ((uint32_t)m <= order) {
      bad = bad || !(u[0] > 0.0);
      // k = -v[1] / u[0] by v_rcp_f64 and two Newton steps (~1e-16 relative) instead of the IEEE division's twenty-odd
      // instructions: this recursion only has to be ACCURATE -- its result is the end of a bracket whose width is at least
      // cert (2 order + 1) (n 2^-53) ~ 1e-9 of r0, and the caller widens the logarithms by 1e-11 on top -- not reproducible
      // bit for bit (nothing of it reaches the stream: DESIGN section 2a)
      double rc0 = __builtin_amdgcn_rcp(u[0]);
      rc0 = __builtin_fma(__builtin_fma(-u[0], rc0, 1.0), rc0, rc0);
      rc0 = __builtin_fma(__builtin_fma(-u[0], rc0, 1.0), rc0, rc0);
      const double k = -v[1] * rc0;
      bad = bad || !(fabs(k) < 1.0);
      growth *= 1.0 + fabs(k);                            // prod (1 + |k_j|) >= ||a||_1 of every predictor on the way
#pragma unroll
      for (int i = 0; i <= P - m; i++) {
        const double t = v[i + 1], ui = u[i];
        u[i] = __builtin_fma(k, t, ui);
        v[i] = __builtin_fma(k, ui, t);
      }
    }
  }
  return (!bad && u[0] > 0.0) ? u[0] : nan;
}

// Levinson-Durbin of one candidate entirely in registers (the recursion of k_blocks_finish, src/SLAPredictor.c:253-328;
// stages and coefficient indices unrolled): o = { r0, parcor[0..order] }.  Same operations in the same order as
// levinson_out, which keeps its vectors in LDS.
template <int P>
__device__ __forceinline__ void levinson_regs(const double* __restrict__ rc, double* __restrict__ o, uint32_t order, uint32_t n)
{
  double r[P + 1], a[P + 1], par[P + 1];
#pragma unroll
  for (int i = 0; i <= P; i++) { r[i] = ((uint32_t)i <= order) ? rc[i] : 0.0; a[i] = 0.0; par[i] = 0.0; }
  if (!(n < order || fabs(r[0]) < (double)FLT_EPSILON)) {
    a[0] = 1.0;
    a[1] = -r[1] / r[0];
    par[1] = r[1] / r[0];
    double e = r[0] + r[1] * a[1];
#pragma unroll
    for (int d = 1; d < P; d++) {
      if ((uint32_t)d < order) {
        double gamma = 0.0;
#pragma unroll
        for (int i = 0; i <= d; i++) { gamma += a[i] * r[d + 1 - i]; }
        gamma /= (-e);
        e = (1.0 - gamma * gamma) * e;
        double nw[P + 1];
#pragma unroll
        for (int i = 1; i <= d; i++) { nw[i] = a[i] + gamma * a[d + 1 - i]; }
#pragma unroll
        for (int i = 1; i <= d; i++) { a[i] = nw[i]; }
        a[0] = 1.0 + gamma * 0.0;
        a[d + 1] = 0.0 + gamma * 1.0;
        par[d + 1] = -gamma;
      }
    }
  }
  o[0] = r[0];
#pragma unroll
  for (int j = 0; j <= P; j++) { if ((uint32_t)j <= order) { o[1 + j] = par[j]; } }
}

// One wave takes `gpw` groups at once when their candidates fit its lanes (a 4096-sample window has 10 candidates: six
// groups per wave instead of one wave with 10 busy lanes per group -- the kernel is bound by instruction issue); a
// group with more than 64 candidates (windows above 8192 samples) takes several passes of one wave.
// LDS: r[lanes][order+1] | a[lanes][order+2] | v[lanes][order+2] (work space of the exact windows' Levinson recursion)
template <int P, int mode>           // P >= order: 16, 32, 48, 64
__global__ __launch_bounds__(64)
void k_search_finish(uint32_t order, uint32_t lags, uint32_t gpw, uint32_t per,
                     const sla_hip_lpc_group* __restrict__ groups, uint32_t num_groups, const sla_hip_lpc_cand* __restrict__ cands,
                     const double* __restrict__ tile_sums, double* __restrict__ out, double exact_limit, double cert,
                     uint32_t* __restrict__ any_exact)
{
  // any_exact (may be NULL): mode 1 sets the word when it meets a group below the limit, mode 2 returns at once while it is
  // still 0 -- loud material wider than 16 bits has no such group, and the mode-2 launch (70 KiB of LDS per workgroup) cost
  // 0.15 ms per C5-120 s launch just to find that out group by group
  if (mode == 2 && any_exact != nullptr && *any_exact == 0u) { return; }
  // mode 0: every group is known to be under the exactness limit (16-bit material): Levinson-Durbin in registers, LDS only
  // holds r; 1: only the groups over the limit (certificate; LDS only holds r: three times the waves per CU at order 48);
  // 2: only the groups below it, Levinson-Durbin with its vectors in LDS.  Material that may have both kinds is launched
  // as 1 + 2.
  extern __shared__ __attribute__((aligned(16))) double lds[];
  __shared__ sla_hip_lpc_group s_g[XF_GROUPS];
  __shared__ double s_energy[XF_GROUPS];
  __shared__ uint32_t s_skip[XF_GROUPS];
  __shared__ uint32_t s_cmax;
  const uint32_t O1 = order + 1, O2 = order + 2;
  const uint32_t lane = threadIdx.x;
  const uint32_t g0 = blockIdx.x * gpw;
  const uint32_t ngr = (num_groups - g0 < gpw) ? (num_groups - g0) : gpw;
  const uint32_t slots = gpw * per;                  // lanes in use, <= 64
  if (lane == 0) { s_cmax = 0; }
  __syncthreads();
  if (lane < ngr) {
    const sla_hip_lpc_group g = groups[g0 + lane];
    const uint32_t ntiles = (g.num_samples + SLA_HIP_XTILE - 1) / SLA_HIP_XTILE;
    const double* ts = tile_sums + (uint64_t)(g0 + lane) * SLA_HIP_XTILES * 2 * lags;
    double energy = 0.0;
    for (uint32_t t = 0; t < ntiles; t++) { energy += ts[(uint64_t)t * 2 * lags]; }
    const bool skip = (mode == 1 && energy < exact_limit) || (mode == 2 && !(energy < exact_limit));
    if (mode == 1 && skip && any_exact != nullptr) { atomicOr(any_exact, 1u); }
    s_g[lane] = g; s_energy[lane] = energy; s_skip[lane] = skip ? 1u : 0u;
    if (!skip) { atomicMax(&s_cmax, g.cand_count); }
  }
  __syncthreads();
  const uint32_t cmax = s_cmax;
  double* r = lds;                                   // [slots][O1]
  double* av = lds + (size_t)slots * O1;             // [slots][O2]
  double* vv = av + (size_t)slots * O2;              // [slots][O2]
  const double inf = __longlong_as_double(0x7FF0000000000000ll);
  for (uint32_t c0 = 0; c0 < cmax; c0 += per) {
    const uint32_t total = ngr * per * O1;
    for (uint32_t q = lane; q < total; q += 64) {
      const uint32_t gl = q / (per * O1), rem = q - gl * per * O1;
      const uint32_t cl = rem / O1, lag = rem - cl * O1, ci = c0 + cl;
      if (ci < s_g[gl].cand_count && !s_skip[gl]) {
        const sla_hip_lpc_cand cd = cands[s_g[gl].cand_first + ci];
        const double* ts = tile_sums + (uint64_t)(g0 + gl) * SLA_HIP_XTILES * 2 * lags;
        const uint32_t end = cd.start + cd.len;
        double sum = 0.0;
        if (lag < cd.len) {
          const uint32_t tl = (end - 1) / SLA_HIP_XTILE;
          for (uint32_t t = cd.start / SLA_HIP_XTILE; t <= tl; t++) { sum += ts[(uint64_t)t * 2 * lags + lag]; }
          sum -= ts[(uint64_t)tl * 2 * lags + lags + lag];
        }
        r[(gl * per + cl) * O1 + lag] = sum;
      }
    }
    __syncthreads();
    const uint32_t gl = lane / per, cl = lane - gl * per, ci = c0 + cl;
    if (gl < ngr && ci < s_g[gl].cand_count && !s_skip[gl]) {
      const sla_hip_lpc_group g = s_g[gl];
      const double energy = s_energy[gl];
      const sla_hip_lpc_cand cd = cands[g.cand_first + ci];
      double* o = out + ((uint64_t)g.slot_first + ci) * O2;
      const double* rc = r + (size_t)lane * O1;
      if (energy < exact_limit) {
        if (mode == 0) { levinson_regs<P>(rc, o, order, cd.len); }
        else if (mode == 2) { levinson_regs<P>(rc, o, order, cd.len); }      // (registers, as mode 0: no work space in LDS)
      } else if (mode != 1 || !(cert > 0.0)) {
        o[0] = __longlong_as_double(0x7FF8000000000000ll);          // flagged: rerun as serial chains
      } else {
        // slot layout of a certified candidate: { r0, width, log2(e_p / r0), 0, .. }
        const double u = 1.1102230246251565e-16;                    // 2^-53
        const double r0 = rc[0];
        double w = inf, lg = 0.0;
        if (cd.len >= order && r0 > 2.0 * (double)FLT_EPSILON) {     // (the reference zeroes the coefficients below FLT_EPSILON, src/SLAPredictor.c:274)
          const double delta = ((double)cd.len * u) * r0 + (48.0 * u) * energy;
          const double d = cert * (double)(2 * order + 1) * delta;
          // one copy of the unrolled recursion, run for both ends of the bracket (inlined copies do not fit the instruction
          // cache).  The value handed to k_plan is the middle of the bracket in the logarithm, which is where the width applies:
          // log2(e_p) of the reference lies in [log2 e_lo, log2 e_hi] = mid +- w.  (A third recursion on the unshifted sums,
          // used as the middle until round 3, bought nothing but a third of this kernel's time.)
          double e2[2], g2[2] = {1.0, 1.0};
#pragma unroll 1
          for (int bk = 0; bk < 2; bk++) {
            const double rb = (bk == 0) ? (r0 + d) : (r0 - d);
            e2[bk] = (bk == 1 && !(r0 - d > (double)FLT_EPSILON)) ? __longlong_as_double(0x7FF8000000000000ll)
                                                                  : schur_error<P>(rc, rb, order, g2[bk]);
          }
          const double e_hi = e2[0], e_lo = e2[1];
          // The factor cert - 1 on top of the summation bounds is what is left for the rounding of the Levinson-Durbin
          // recursion itself (the reference's run): to first order it enters stage m through the sum num_m = sum a_i r_(m-i)
          // with at most (m + 2) 2^-53 ||a^(m-1)||_1 r0, i.e. like an autocorrelation error of that size -- covered by
          // d = cert (2 order + 1) delta as long as (order + 2) ||a||_1 2^-53 r0 <= (cert - 1) (2 order + 1) delta.
          // ||a||_1 <= prod (1 + |k_j|) of the more pessimistic bracket end; a candidate beyond that is not certified.
          const double gmax = fmax(g2[0], g2[1]);
          const bool rounding_covered = ((double)(order + 2) * gmax * u * r0 <= (cert - 1.0) * (double)(2 * order + 1) * delta);
          if (rounding_covered && e_lo > 0.0 && e_hi >= e_lo && e_hi < inf) {
            const double lh = log2(e_hi / r0), ll = log2(e_lo / r0);
            w = 0.5 * (lh - ll) * 1.000001 + 1e-11;               // (device log2: a few ulp)
            lg = 0.5 * (lh + ll);
          }
        }
        o[0] = r0;
        o[1] = (w == w) ? w : inf;
        o[2] = lg;
        for (uint32_t k = 2; k <= order; k++) { o[1 + k] = 0.0; }
      }
    }
    __syncthreads();
  }
}

// ---------------------------------------------------------------------------------------------
// k_plan: the partition search's scalar tail on the device -- code-length estimate per candidate
// (src/SLAPredictor.c:416-468), adjacency matrix, Dijkstra (:1521-1581) -- so that the host does not have to
// fetch (order+2) doubles per candidate and evaluate 18 logarithms for each.  The reference's decisions hang
// on glibc's log(); the device's log() may differ from it in the last bits, so a result is only accepted
// when it provably does not depend on them: every comparison Dijkstra makes (minimum selection, relaxation)
// and every branch of the estimate must be decided by more than PLAN_MARGIN, which is > 1000x the largest
// possible discrepancy of a path cost (a few 1e-8 for 8 channels x 16384 samples x 16 edges).  Anything
// closer, non-finite or out of range is flagged and the host redoes that super-frame exactly as before.
// One wave per super-frame: lanes = candidates for the costs, lanes = nodes for the relaxation.
// ---------------------------------------------------------------------------------------------
#define PLAN_NODES 17            // 16384 / 1024 + 1
#define PLAN_MARGIN 1e-4
#define PLAN_BIG 16777216.0      // SLAOPTIMALENCODEESTIMATOR_DIJKSTRA_BIGWEIGHT

// `width`: half width of log2(e_p) the candidate carries (0: its sums are the reference's, bit for bit)
__device__ __forceinline__ double plan_code_length(double sumsq, uint32_t n, uint32_t bps, const double* __restrict__ parcor,
                                                   uint32_t order, double width, bool& sure)
{
  const double l2e = 1.4426950408889634;                        // src/SLAUtility.c:442-447
  double power = sumsq * ldexp(1.0, (int)(2 * (bps - 1)));
  if (fabs(power) <= (double)FLT_MIN) { if (width > 0.0) { sure = false; } return 0.0; }
  power = log(power) * l2e - log((double)n) * l2e;
  double gain = 0.0;
  if (width != 0.0) { gain = parcor[1]; }                        // certified candidate: log2(e_p / r0) itself (k_search_finish)
  else {
    // sum of log(1 - k^2), eight factors per logarithm: the value only has to agree with the host's sum of single
    // logarithms to within the margin every comparison must clear (1e-4 bytes; this differs by ~1e-11), and a product
    // of eight factors >= 2^-53 cannot underflow
    double prod = 1.0;
    for (uint32_t ord = 1; ord <= order; ord++) {
      prod *= 1.0 - parcor[ord] * parcor[ord];
      if ((ord & 7u) == 0 || ord == order) { gain += log(prod) * l2e; prod = 1.0; }
    }
  }
  double len = 1.9426950408889634 + 0.5 * (power + gain);
  len /= 8;
  if (!(fabs(len) > 1e-9 + width / 16.0)) { sure = false; }     // too close to the clamp (or NaN)
  return (len <= 0) ? 0.125 : len;
}

__global__ __launch_bounds__(256)
void k_plan(const sla_hip_lpc_group* __restrict__ groups, uint32_t num_sf, uint32_t nch, uint32_t order, uint32_t bps,
            const sla_hip_lpc_cand* __restrict__ cands, double* __restrict__ lpc_out,
            uint32_t* __restrict__ parts, uint32_t* __restrict__ nparts, uint32_t* __restrict__ status, double margin)
{
  __shared__ double s_adj[4][PLAN_NODES * PLAN_NODES];
  __shared__ uint32_t s_path[4][PLAN_NODES];
  const uint32_t lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const uint32_t sf = blockIdx.x * 4 + wv;
  if (sf >= num_sf) { return; }
  const sla_hip_lpc_group g = groups[(uint64_t)sf * nch];       // channel 0 of the super-frame: all its candidates
  const uint32_t window = g.num_samples, O2 = order + 2;
  const uint32_t nodes = (window + SLA_HIP_XTILE - 1) / SLA_HIP_XTILE + 1;
  double* adj = s_adj[wv];
  bool sure = (nodes <= PLAN_NODES), inexact = false;
  double wmax = 0.0;
  if (!sure) { if (lane == 0) { status[sf] = 1; nparts[sf] = 0; } return; }
  for (uint32_t q = lane; q < nodes * nodes; q += 64) { adj[q] = PLAN_BIG; }
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  for (uint32_t k = lane; k < g.cand_count; k += 64) {
    const sla_hip_lpc_cand cd = cands[g.cand_first + k];
    const uint32_t i = cd.start / SLA_HIP_XTILE, j = (cd.start + cd.len + SLA_HIP_XTILE - 1) / SLA_HIP_XTILE;
    double wedge = 0.0, est = 0.0;
    for (uint32_t ch = 0; ch < nch; ch++) {
      const double* o = lpc_out + ((uint64_t)g.slot_first + (uint64_t)ch * g.cand_count + k) * O2;
      const double width = o[1];                   // parcor[0]: 0, or the half width of log2(e_p) of a certified candidate
      est += cd.len * plan_code_length(o[0], cd.len, bps, o + 1, order, width, sure);
      wedge += cd.len * width / 16.0;              // bytes: length = n/8 * (const + log2(e_p * scale / n) / 2)
      if (width != 0.0) { inexact = true; }
    }
    if (!(wedge < PLAN_BIG / 2)) { sure = false; wedge = 0.0; }
    wmax = (wedge > wmax) ? wedge : wmax;
    est += 50.0;                                   // SLAOPTIMALENCODEESTIMATOR_ESTIMATE_BLOCK_SIZE
    est += 300.0;                                  // ..._LONGPATH_PENALTY
    if (!(fabs(est) < PLAN_BIG / 2)) { sure = false; }          // NaN (also the "rerun as serial chains" flag), inf, absurd
    if (i < nodes && j < nodes && j > i) { adj[i * nodes + j] = est; } else { sure = false; }
  }
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");

  // a path has at most nodes-1 edges, each known to +-wmax: two path costs compare safely beyond 2*(nodes-1)*wmax
  for (int off = 32; off > 0; off >>= 1) { const double o = __shfl_xor(wmax, off); wmax = (o > wmax) ? o : wmax; }
  margin += 2.0 * (double)(nodes - 1) * wmax;
  inexact = (__ballot(inexact) != 0ull);
  // Dijkstra, lane = node: first-minimum selection, strict-improvement relaxation (as slai_shortest_path)
  double cost = (lane == 0) ? 0.0 : PLAN_BIG;
  bool done = false, reached = false;
  uint32_t pred = 0xFFFFFFFFu;
  const double inf = __longlong_as_double(0x7FF0000000000000ll);
  for (uint32_t round = 0; round <= nodes; round++) {
    const double v = (lane < nodes && !done && cost < PLAN_BIG) ? cost : inf;
    double best = v;
    for (int off = 32; off > 0; off >>= 1) { const double o = __shfl_xor(best, off); best = (o < best) ? o : best; }
    if (!(best < PLAN_BIG)) { break; }                         // the reference would not terminate: the host reports it
    const unsigned long long at = __ballot(v == best);
    const uint32_t cur = (uint32_t)__builtin_ctzll(at);
    double second = (lane == cur) ? inf : v;
    for (int off = 32; off > 0; off >>= 1) { const double o = __shfl_xor(second, off); second = (o < second) ? o : second; }
    if (second - best < margin) { sure = false; }
    if (cur == nodes - 1) { reached = true; break; }
    if (lane < nodes) {
      const double a = adj[cur * nodes + lane];
      if (a < PLAN_BIG) {                                      // BIG + x never improves a cost (all costs <= BIG)
        const double via = a + best;
        if (fabs(cost - via) < margin) { sure = false; }
        if (cost > via) { cost = via; pred = cur; }
      }
    }
    if (lane == cur) { done = true; }
  }
  if (lane < nodes) { s_path[wv][lane] = pred; }
  const bool all_sure = (__ballot(!sure) == 0ull);
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  if (lane == 0) {
    uint32_t count = 0, node = nodes - 1;
    bool ok = reached && all_sure;
    while (ok && node != 0) {
      const uint32_t pr = s_path[wv][node];
      if (pr >= node) { ok = false; break; }
      count++; node = pr;
    }
    if (ok) {
      node = nodes - 1;
      for (uint32_t q = 0; q < count; q++) {
        const uint32_t pr = s_path[wv][node];
        const uint32_t off = pr * SLA_HIP_XTILE;
        uint32_t len = (node - pr) * SLA_HIP_XTILE;
        if (len > window - off) { len = window - off; }
        parts[(uint64_t)sf * PLAN_NODES + (count - q - 1)] = len;
        node = pr;
      }
    }
    nparts[sf] = ok ? count : 0;
    status[sf] = ok ? 0u : (inexact ? 2u : 1u);
    s_path[wv][0] = (ok || !inexact) ? 0u : 1u;
  }
  // a super-frame whose tile sums were only certified, not exact, and that did not certify: flag every candidate for
  // the serial-chain rerun (sla_hip_launch_lpc_rerun looks at r[0] of a group's first slot)
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  if (s_path[wv][0] != 0u) {
    for (uint32_t q = lane; q < g.cand_count * nch; q += 64) {
      lpc_out[((uint64_t)g.slot_first + q) * O2] = __longlong_as_double(0x7FF8000000000000ll);
    }
  }
}

// ---------------------------------------------------------------------------------------------
// k_expand: the block table of one run of super-frames from k_plan's partitions (see sla_hip_launch_expand).  One
// workgroup: every thread owns a run of consecutive super-frames, counts their blocks, a workgroup scan numbers them,
// and each thread writes the descriptors of its own blocks.  A ten-minute mono file is 7 k super-frames, an hour of
// stereo 42 k: microseconds, against the 0.15 - 0.2 ms the host needs to fetch the partitions, build the same tables
// and upload them while the device waits.
// ---------------------------------------------------------------------------------------------
#define EXPAND_THREADS 1024
typedef uint32_t ex_u32x4 __attribute__((ext_vector_type(4), aligned(4)));
typedef uint32_t ex_u32x2 __attribute__((ext_vector_type(2), aligned(4)));
#define EXPAND_PER 8                               // super-frames per thread and tile
#define EXPAND_TILE (EXPAND_THREADS * EXPAND_PER)
#define EXPAND_MAX_WINDOWS 256
#define EXPAND_MAX_LEN 16384u                      // longest block the device analyses (the window lives in LDS)
// Two kernels.  k_expand_scan (ONE workgroup) numbers the blocks: super-frames in tiles of 8192, thread t owning t,
// t + 1024, .. so that the loads of a thread's eight super-frames are independent and coalesced across the workgroup, the
// counts in LDS, where the scan runs; it writes two prefix words per super-frame, validates every block length against the
// window list and publishes the counts.  k_expand_write (one thread per super-frame, the whole device) writes the
// descriptors.  (First version: one workgroup did everything, every thread walking a run of consecutive super-frames
// twice -- 60 us for a ten-minute mono file, 300 us for an hour of stereo, most of it one CU pushing a megabyte of
// scattered 40-byte records per tile through its store path.)
__device__ __forceinline__ uint32_t expand_window(const uint32_t* s_wlen, const uint32_t* s_woff, uint32_t num_win, uint32_t len)
{
  for (uint32_t q = 0; q < num_win; q++) { if (s_wlen[q] == len) { return s_woff[q]; } }
  return SLA_HIP_NO_WINDOW;
}

__global__ __launch_bounds__(EXPAND_THREADS)
void k_expand_scan(const sla_hip_superframe* __restrict__ sf, uint32_t num_sf, const uint32_t* __restrict__ parts,
                   const uint32_t* __restrict__ nparts, const uint32_t* __restrict__ status, uint32_t nch,
                   const uint32_t* __restrict__ win_len, uint32_t num_win,
                   uint32_t* __restrict__ run, uint32_t* __restrict__ prefix, uint32_t capacity,
                   volatile uint32_t* counts, uint32_t sequence)
{
  __shared__ uint32_t s_has[EXPAND_MAX_LEN / 32 + 1];             // bit per block length: its window table exists
  __shared__ uint32_t s_cb[EXPAND_TILE], s_cl[EXPAND_TILE];       // per super-frame of the tile: blocks, blocks of live super-frames
  __shared__ uint32_t s_wb[EXPAND_THREADS / 64], s_wl[EXPAND_THREADS / 64];
  __shared__ uint32_t s_bad;
  const uint32_t t = threadIdx.x, lane = t & 63, wv = t >> 6;
  if (t == 0) { s_bad = (run[2] != 0u || num_win > EXPAND_MAX_WINDOWS) ? 1u : 0u; }
  for (uint32_t i = t; i < EXPAND_MAX_LEN / 32 + 1; i += EXPAND_THREADS) { s_has[i] = 0u; }
  __syncthreads();
  for (uint32_t i = t; i < num_win && i < EXPAND_MAX_WINDOWS; i += EXPAND_THREADS) {
    const uint32_t len = win_len[i];
    if (len != 0u && len <= EXPAND_MAX_LEN) { atomicOr(&s_has[len >> 5], 1u << (len & 31u)); }
  }
  const uint32_t run_b = run[0], run_g = run[1];
  uint32_t base_b = 0, base_l = 0;                 // blocks / live blocks of the tiles before this one
  bool bad = false;
  __syncthreads();
  for (uint32_t t0 = 0; t0 < num_sf; t0 += EXPAND_TILE) {
    // my super-frames' plan rows and first block lengths: two rounds of loads, each round's eight requested before the
    // first is used (a super-frame is one block more often than not: its length rides along)
    uint32_t live[EXPAND_PER], np[EXPAND_PER], st[EXPAND_PER], len0[EXPAND_PER];
#pragma unroll
    for (int k = 0; k < EXPAND_PER; k++) {
      const uint32_t i = t0 + (uint32_t)k * EXPAND_THREADS + t;
      live[k] = (i < num_sf) ? sf[i].live : SLA_HIP_NOT_LIVE;
    }
#pragma unroll
    for (int k = 0; k < EXPAND_PER; k++) {
      const bool l = (live[k] != SLA_HIP_NOT_LIVE);
      np[k] = l ? nparts[live[k]] : 0u;
      st[k] = l ? status[live[k]] : 0u;
      len0[k] = l ? parts[(uint64_t)live[k] * PLAN_NODES] : 0u;
    }
#pragma unroll
    for (int k = 0; k < EXPAND_PER; k++) {
      const uint32_t i = t0 + (uint32_t)k * EXPAND_THREADS + t;
      const bool l = (live[k] != SLA_HIP_NOT_LIVE);
      if (l && (st[k] != 0u || np[k] == 0u || np[k] > PLAN_NODES)) { bad = true; np[k] = 0; }
      for (uint32_t p = 0; l && p < np[k]; p++) {          // every block length must have its window table
        const uint32_t len = (p == 0) ? len0[k] : parts[(uint64_t)live[k] * PLAN_NODES + p];
        if (len == 0u || len > EXPAND_MAX_LEN || ((s_has[len >> 5] >> (len & 31u)) & 1u) == 0u) { bad = true; }
      }
      s_cb[(uint32_t)k * EXPAND_THREADS + t] = (i < num_sf) ? (l ? np[k] : 1u) : 0u;      // a SILENT super-frame is one block
      s_cl[(uint32_t)k * EXPAND_THREADS + t] = l ? np[k] : 0u;
    }
    __syncthreads();
    // exclusive scan over the tile, in LDS: thread t sums entries [8 t, 8 t + 8), wave scan, wave totals
    uint32_t cb[EXPAND_PER], cl[EXPAND_PER], my_b = 0, my_l = 0;
#pragma unroll
    for (int k = 0; k < EXPAND_PER; k++) { cb[k] = s_cb[t * EXPAND_PER + k]; cl[k] = s_cl[t * EXPAND_PER + k]; my_b += cb[k]; my_l += cl[k]; }
    uint32_t inc_b = my_b, inc_l = my_l;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
      const uint32_t ob = (uint32_t)__shfl_up((int)inc_b, off), ol = (uint32_t)__shfl_up((int)inc_l, off);
      if (lane >= (uint32_t)off) { inc_b += ob; inc_l += ol; }
    }
    if (lane == 63) { s_wb[wv] = inc_b; s_wl[wv] = inc_l; }
    __syncthreads();
    uint32_t off_b = run_b + base_b + (inc_b - my_b), off_l = base_l + (inc_l - my_l), tile_b = 0, tile_l = 0;
    for (uint32_t w = 0; w < EXPAND_THREADS / 64; w++) {
      const uint32_t vb = s_wb[w], vl = s_wl[w];
      if (w < wv) { off_b += vb; off_l += vl; }
      tile_b += vb; tile_l += vl;
    }
    // block number and live-block number of each of my eight consecutive entries: 64 contiguous bytes per thread
#pragma unroll
    for (int k = 0; k < EXPAND_PER; k++) {
      const uint32_t i = t0 + t * EXPAND_PER + (uint32_t)k;
      if (i < num_sf) { ex_u32x2 w; w.x = off_b; w.y = off_l; *reinterpret_cast<ex_u32x2*>(prefix + 2 * (uint64_t)i) = w; }
      off_b += cb[k]; off_l += cl[k];
    }
    base_b += tile_b; base_l += tile_l;
    __syncthreads();                                   // s_cb / s_cl are rewritten by the next tile
  }
  if (bad) { s_bad = 1u; }
  __syncthreads();
  if (t == 0) {
    const bool ok = (s_bad == 0u) && ((uint64_t)run_g + (uint64_t)base_l * nch <= (uint64_t)capacity);
    // run[3]: first group of this run, for k_expand_write (run[1] moves on)
    run[3] = run_g;
    if (ok) { run[0] = run_b + base_b; run[1] = run_g + base_l * nch; } else { run[2] = 1u; }
    counts[0] = ok ? base_b : 0u; counts[1] = ok ? base_l * nch : 0u; counts[2] = ok ? 1u : 0u;
    __threadfence_system();
    counts[3] = sequence;
    __threadfence_system();
  }
}

__global__ __launch_bounds__(256)
void k_expand_write(const sla_hip_superframe* __restrict__ sf, uint32_t num_sf, const uint32_t* __restrict__ parts,
                    const uint32_t* __restrict__ nparts, uint32_t nch, uint32_t int_shift,
                    const uint32_t* __restrict__ win_len, const uint32_t* __restrict__ win_off, uint32_t num_win,
                    const uint32_t* __restrict__ run, const uint32_t* __restrict__ prefix,
                    sla_hip_lpc_group* __restrict__ groups, sla_hip_lpc_cand* __restrict__ cands, sla_hip_acf_job* __restrict__ acf_jobs)
{
  __shared__ uint32_t s_wlen[EXPAND_MAX_WINDOWS], s_woff[EXPAND_MAX_WINDOWS];
  if (run[2] != 0u) { return; }                        // the scan found this run (or an earlier one) unfit for device tables
  for (uint32_t i = threadIdx.x; i < num_win && i < EXPAND_MAX_WINDOWS; i += blockDim.x) { s_wlen[i] = win_len[i]; s_woff[i] = win_off[i]; }
  __syncthreads();
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= num_sf) { return; }
  const sla_hip_superframe f = sf[i];
  if (f.live == SLA_HIP_NOT_LIVE) { return; }
  const uint32_t run_g = run[3];
  const uint32_t np = nparts[f.live];
  uint32_t b = prefix[2 * (uint64_t)i], l = prefix[2 * (uint64_t)i + 1], at = f.start;
  for (uint32_t p = 0; p < np; p++, b++, l++) {
    const uint32_t len = parts[(uint64_t)f.live * PLAN_NODES + p];
    const uint32_t woff = expand_window(s_wlen, s_woff, num_win, len);
    for (uint32_t ch = 0; ch < nch; ch++) {
      const uint32_t g = run_g + l * nch + ch;
      // 64 bytes per (block, channel), as five wide stores
      static_assert(sizeof(sla_hip_lpc_group) == 40 && sizeof(sla_hip_lpc_cand) == 8 && sizeof(sla_hip_acf_job) == 16, "descriptor layout");
      ex_u32x4 w0, w1; ex_u32x2 w2, wc; ex_u32x4 wa;
      w0.x = at; w0.y = 0u; w0.z = len; w0.w = ch;                       // pcm_off (u64), num_samples, channel
      w1.x = woff; w1.y = int_shift; w1.z = g; w1.w = 1u;                // win_off, int_shift, cand_first, cand_count
      w2.x = b * nch + ch; w2.y = 0u;                                    // slot_first, pad_
      uint32_t* gp = reinterpret_cast<uint32_t*>(groups + g);
      *reinterpret_cast<ex_u32x4*>(gp) = w0; *reinterpret_cast<ex_u32x4*>(gp + 4) = w1; *reinterpret_cast<ex_u32x2*>(gp + 8) = w2;
      wc.x = 0u; wc.y = len;
      *reinterpret_cast<ex_u32x2*>(cands + g) = wc;
      wa.x = at; wa.y = 0u; wa.z = len; wa.w = ch;                       // blk_off (u64), blk_len, channel
      *reinterpret_cast<ex_u32x4*>(acf_jobs + g) = wa;
    }
    at += len;
  }
}

// ---------------------------------------------------------------------------------------------
// k_lattice: one wave per chunk; every lane keeps T consecutive samples of the forward and
// backward prediction errors in registers, stage m needs b_{m-1}[n-1] of the previous lane
// (one DPP-able shuffle per stage).  The first H lanes re-compute `order` samples of history
// (the lattice is feed-forward: output n depends on inputs n-order..n only), so chunks are
// independent and need no LDS and no barrier.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256, 8)
void k_lattice(const int32_t* __restrict__ pcm, uint64_t stride, uint32_t ms, uint32_t order,
               const sla_hip_lattice_chunk* __restrict__ chunks, uint32_t num_chunks,
               const int32_t* __restrict__ kint, int32_t* __restrict__ residual, unsigned long long* span, uint32_t flags)
{
  __shared__ int32_t s_tile[4][LAT_TILE_WORDS];
  const uint32_t lane = threadIdx.x & 63;
  const uint32_t wv = (uint32_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));      // wave-uniform: descriptors and coefficients by scalar loads
  const uint32_t cid = blockIdx.x * 4 + wv;
  if (cid >= num_chunks) { return; }
  span_begin(span);
  const sla_hip_lattice_chunk ck = chunks[cid];
  lattice_chunk_wave_lds(pcm, stride, ms, order, ck.blk_off, ck.blk_len, ck.chunk_start, ck.count, ck.channel, ck.int_shift,
                         kint + (uint64_t)ck.slot * (order + 1), residual, lane, (flags & 1u) != 0, s_tile[wv], (flags & 2u) != 0);
  span_end(span);
}

// The same lattice waves, derived from the block descriptors themselves: wave = (group, chunk index), `cpg` waves reserved
// per group (enough for the longest block), the ones past a block's end return at once.  Saves the host a descriptor per
// ~900 samples (4.5 MB for ten minutes of stereo: building and uploading them took longer than k_lpc_blocks runs).
__global__ __launch_bounds__(256, 8)
void k_lattice_groups(const int32_t* __restrict__ pcm, uint64_t stride, uint32_t ms, uint32_t order,
                      const sla_hip_lpc_group* __restrict__ groups, uint32_t num_groups, uint32_t cpg,
                      const int32_t* __restrict__ kint, int32_t* __restrict__ residual, unsigned long long* span, uint32_t flags)
{
  __shared__ int32_t s_tile[4][LAT_TILE_WORDS];
  const uint32_t lane = threadIdx.x & 63;
  const uint32_t wv = (uint32_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const uint32_t w = blockIdx.x * 4 + wv;
  const uint32_t gi = w / cpg, c = w - gi * cpg;
  if (gi >= num_groups) { return; }
  span_begin(span);
  const sla_hip_lpc_group g = groups[gi];
  const uint32_t per = (SLA_WAVE - (order + LAT_T - 1) / LAT_T) * LAT_T;
  const uint32_t at = c * per;
  if (at < g.num_samples) {
    lattice_chunk_wave_lds(pcm, stride, ms, order, g.pcm_off, g.num_samples, at, (g.num_samples - at < per) ? (g.num_samples - at) : per,
                           g.channel, g.int_shift, kint + (uint64_t)g.slot_first * (order + 1), residual, lane, false, s_tile[wv], (flags & 2u) != 0);
  }
  span_end(span);
}

// ---------------------------------------------------------------------------------------------
// Pre-emphasis as its own pass (per-call SLAEmphasisFilter API; the pipeline fuses it into k_lpc_blocks / k_lattice):
// y[n] = x[n] - ((x[n-1] * (2^s - 1)) >> s) with x[-1] = prev      src/SLAPredictor.c:1741-1765, 1794-1813
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256)
void k_emphasis_i32(const int32_t* __restrict__ in, int32_t* __restrict__ out, uint32_t n, int32_t prev, uint32_t shift)
{
  const uint32_t i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n) { return; }
  const int32_t p = (i == 0) ? prev : in[i - 1];
  const int32_t coef = (int32_t)((1u << shift) - 1u);
  out[i] = (int32_t)((uint32_t)in[i] - (uint32_t)((int32_t)((uint32_t)p * (uint32_t)coef) >> shift));
}

__global__ __launch_bounds__(256)
void k_emphasis_f64(const double* __restrict__ in, double* __restrict__ out, uint32_t n, uint32_t shift)
{
  const uint32_t i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n) { return; }
  const double coef = (ldexp(1.0, (int)shift) - 1.0) * ldexp(1.0, -(int)shift);
  const double p = (i == 0) ? 0.0 : in[i - 1];
  out[i] = in[i] - p * coef;
}

// ---------------------------------------------------------------------------------------------
// Tail stage: long-term filter -> sign-log LMS -> folded sum (src/SLAPredictor.c:1031-1119, 1202-1331; src/SLACoder.c:361-385).
// The LMS is serial in time (every sample updates all 2 * ORDER coefficients from the error it just produced), so the
// parallelism inside one (block, channel) job exists only ACROSS THE TAPS: a few lanes own one job, each K taps of the input
// history and K of the prediction history; per sample the lanes' products meet in a DPP sum (wrapping int adds are
// associative, so the tree equals the reference's serial sum), the error and the step are computed redundantly by every
// lane of the job, and the histories move on by one DPP shift.  (Rounds 1 - 3 kept three earlier layouts -- 2 * ORDER
// lanes, ORDER lanes, ONE lane per job -- behind an option; k_tailk beat all of them at every job count)
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ int32_t sgn(int32_t v)          // clamp to [-1, 1] = sign, one v_med3_i32
{
  int32_t r;
  asm("v_med3_i32 %0, %1, -1, 1" : "=v"(r) : "v"(v));
  return r;
}

__device__ __forceinline__ int32_t mad24(int32_t a, int32_t b, int32_t c)      // a, b within 24 bits; c + a*b wraps like int32
{
  int32_t r;
  asm("v_mad_i32_i24 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
  return r;
}

// all lanes of the permutation are valid, so old = 0 / bound_ctrl lets the add absorb the DPP read
template <int CTRL>
__device__ __forceinline__ uint32_t dpp_u32(uint32_t x)
{
  return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, CTRL, 0xF, 0xF, true);
}

// ---------------------------------------------------------------------------------------------
// k_tailk: the same stage with K taps of each history per lane (2 K coefficients), ORDER / K lanes per job -- for files
// with so many (block, channel) jobs that what counts is instructions per job and sample, not the length of one job's
// chain.  Measured (tests/tools/ubench_int.hip, one wave on a SIMD): an instruction that needs the result of a recent one
// issues 3.3 ns after it, an independent one 1.7 - 1.9 ns, a dependent DPP step with its wait states 6.8 ns -- and a
// 32-bit v_mul_lo costs what an add costs.  So a lone wave pays for every instruction of the per-sample loop (k_tail:
// 22 + 4 wait states = 105 ns per sample, which is the whole C2 launch), two or more waves per SIMD fill each other's
// stalls, and then the work per sample is what is left to cut: everything that is computed once per job and sample
// (the sum's last steps, error, logarithm, step, history shift: ~20 instructions) is shared by 64 / (ORDER / K) jobs
// of a wave instead of 4 (k_tail) or 8 (k_tail2), against 4 K instructions more for the lane's own products and
// updates.  ORDER 8: k_tail2 ~30 instructions per sample for 8 jobs, K = 2 ~37 for 16, K = 4 ~53 for 32.
//
// Layout.  The jobs of a DPP row (16 lanes) are interleaved: lane r of the row is lane r / JPR of job r % JPR (JPR = jobs
// per row), so a shift of the history to the job's next lane is row_shr:JPR for all jobs at once -- the first lane of
// every job has no source inside the row and keeps `old`, which is the new input / prediction: no select -- and the
// job's sum is log2(lanes) row rotations.  A lane keeps its K history values in place: at step u of the unrolled block
// the tap of age a sits in slot (a - u) mod K, the slot of the value that leaves for the next lane takes the one that
// arrives, and the signs are kept beside the values (one sign per arriving value instead of one per tap and sample).
// ---------------------------------------------------------------------------------------------
template <int LPJ>
__device__ __forceinline__ uint32_t job_sum(uint32_t x)          // over the lanes r, r + JPR, r + 2 JPR, .. of a row
{
  x += dpp_u32<0x128>(x);                                  // row_ror:8
  if (LPJ >= 4) { x += dpp_u32<0x124>(x); }                // row_ror:4
  if (LPJ >= 8) { x += dpp_u32<0x122>(x); }                // row_ror:2
  if (LPJ >= 16) { x += dpp_u32<0x121>(x); }               // row_ror:1
  return x;
}

// Samples cross global memory 32 at a time per job (TAILK_BLK): the job's lanes each move a run of 32 / lanes consecutive
// samples as 16-byte accesses, so a job touches each of its cache lines once.  (Per-sample accesses of 2 - 4 lanes per
// job made a wave's load touch 16 - 32 lines for 8 bytes each: the first version of this kernel was bound by those
// line fetches and lost to k_tail2 -- C3 60 min 4.6 against 3.6 ms with four taps per lane.)
#define TAILK_BLK 32
typedef int32_t tk_i32x4 __attribute__((ext_vector_type(4), aligned(4)));
typedef int32_t tk_i32x2 __attribute__((ext_vector_type(2), aligned(4)));

template <int N>
__device__ __forceinline__ void tk_load_run(const int32_t* __restrict__ p, int32_t (&v)[N])      // N consecutive words, 4-byte aligned
{
  if constexpr (N % 4 == 0) {
#pragma unroll
    for (int q = 0; q < N / 4; q++) {
      const tk_i32x4 x = *reinterpret_cast<const tk_i32x4*>(p + 4 * q);
      v[4 * q] = x.x; v[4 * q + 1] = x.y; v[4 * q + 2] = x.z; v[4 * q + 3] = x.w;
    }
  } else {
#pragma unroll
    for (int q = 0; q < N / 2; q++) {
      const tk_i32x2 x = *reinterpret_cast<const tk_i32x2*>(p + 2 * q);
      v[2 * q] = x.x; v[2 * q + 1] = x.y;
    }
  }
}
template <int N>
__device__ __forceinline__ void tk_store_run(int32_t* __restrict__ p, const int32_t (&v)[N])
{
  if constexpr (N % 4 == 0) {
#pragma unroll
    for (int q = 0; q < N / 4; q++) {
      tk_i32x4 x; x.x = v[4 * q]; x.y = v[4 * q + 1]; x.z = v[4 * q + 2]; x.w = v[4 * q + 3];
      *reinterpret_cast<tk_i32x4*>(p + 4 * q) = x;
    }
  } else {
#pragma unroll
    for (int q = 0; q < N / 2; q++) {
      tk_i32x2 x; x.x = v[2 * q]; x.y = v[2 * q + 1];
      *reinterpret_cast<tk_i32x2*>(p + 2 * q) = x;
    }
  }
}

template <int ORDER, int K, bool FIRST>
__device__ __forceinline__ void tailk_block(const int32_t (&vm)[TAILK_BLK * K / ORDER], int32_t (&em)[TAILK_BLK * K / ORDER],
                                            const int (&addr)[ORDER / K], uint32_t L,
                                            int32_t (&ca)[K], int32_t (&cb)[K], int32_t (&ha)[K], int32_t (&hb)[K],
                                            int32_t (&sa)[K], int32_t (&sb)[K])
{
  constexpr int LPJ = ORDER / K, JPR = 16 / LPJ, SPL = TAILK_BLK / LPJ;
  constexpr int SHR = 0x110 + JPR;                         // row_shr:JPR
#pragma unroll
  for (int u0 = 0; u0 < TAILK_BLK; u0 += 8) {
    // sample u of the block was fetched by the job's lane u / SPL as its element u % SPL; eight at a time, all requested
    // before the first is used, so that no step of the chain waits for the LDS crossbar
    int32_t vs[8];
#pragma unroll
    for (int w = 0; w < 8; w++) { vs[w] = __builtin_amdgcn_ds_bpermute(addr[(u0 + w) / SPL], vm[(u0 + w) % SPL]); }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int w = 0; w < 8; w++) {
      const int u = u0 + w;
      const int32_t v = vs[w];
      int32_t e, ph;
      if (FIRST && u < ORDER) {
        e = v; ph = v;                                     // the first ORDER samples only prime both histories
      } else {
        uint32_t acc = 0;
#pragma unroll
        for (int a = 0; a < K; a++) {
          const int slot = ((a - u) % K + K) % K;
          acc += (uint32_t)ca[a] * (uint32_t)ha[slot];
          acc += (uint32_t)cb[a] * (uint32_t)hb[slot];
        }
        const uint32_t sum = job_sum<LPJ>(acc) + (1u << 9);
        const int32_t p = (int32_t)sum >> 10;
        e = (int32_t)((uint32_t)v - (uint32_t)p);
        const int32_t ne = (int32_t)((uint32_t)p - (uint32_t)v);
        const uint32_t mag = (uint32_t)max(e, ne);
        // step = ceil(log2(|e| + 1)) >> 1 (step table src/SLAPredictor.c:123-144), times sign(e).  v_ffbh_u32 answers -1 for
        // zero, which would make the step 16 -- times sign(0) = 0: no special case (the portable __clz spelling cost five
        // instructions on this chain, this one three)
        uint32_t lead;
        asm("v_ffbh_u32 %0, %1" : "=v"(lead) : "v"(mag));
        const int32_t g = __mul24(sgn(e), (int32_t)((32u - lead) >> 1));
#pragma unroll
        for (int a = 0; a < K; a++) {
          const int slot = ((a - u) % K + K) % K;
          ca[a] = mad24(g, sa[slot], ca[a]);
          cb[a] = mad24(g, sb[slot], cb[a]);
        }
        ph = p;
      }
      const int put = (K - 1 - (u % K));                   // the slot of the lane's oldest value takes the arriving one
      ha[put] = (int32_t)__builtin_amdgcn_update_dpp(v, ha[put], SHR, 0xF, 0xF, false);
      hb[put] = (int32_t)__builtin_amdgcn_update_dpp(ph, hb[put], SHR, 0xF, 0xF, false);
      sa[put] = sgn(ha[put]);
      sb[put] = sgn(hb[put]);
      em[u % SPL] = (L == (uint32_t)(u / SPL)) ? e : em[u % SPL];
    }
  }
}

template <int ORDER, int K>
__global__ __launch_bounds__(256)
void k_tailk(const int32_t* __restrict__ res_in, int32_t* __restrict__ res_out, uint64_t stride,
             const sla_hip_tail_job* __restrict__ jobs, uint32_t num_jobs, uint32_t ntaps,
             uint64_t* __restrict__ fold_sum, unsigned long long* span, uint32_t stage_flags)
{
  span_begin(span);
  constexpr int LPJ = ORDER / K;               // lanes per job: 2 .. 16
  constexpr int JPR = 16 / LPJ;                // jobs per DPP row
  constexpr int JPW = 4 * JPR;                 // jobs per wave
  constexpr int SPL = TAILK_BLK / LPJ;         // consecutive samples a lane moves per block of 32: 2 .. 16
  static_assert(ORDER % K == 0 && LPJ >= 2 && LPJ <= 16 && (LPJ & (LPJ - 1)) == 0 && TAILK_BLK % ORDER == 0, "lanes per job");
  const uint32_t lane = threadIdx.x & 63;
  const uint32_t row = lane >> 4, r = lane & 15u;
  const uint32_t jr = r % JPR, L = r / JPR;
  const uint32_t wave = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  const uint32_t j = wave * JPW + row * JPR + jr;
  const bool have = (j < num_jobs);
  const sla_hip_tail_job job = jobs[have ? j : 0];
  const uint32_t n = have ? job.blk_len : 0;
  const int32_t* in = res_in + (uint64_t)job.channel * stride + job.blk_off;
  int32_t* out = res_out + (uint64_t)job.channel * stride + job.blk_off;
  const uint32_t delay = job.pitch + (ntaps >> 1);
  const bool use_ltm = (job.pitch >= 3);
  const bool pass = (n < (uint32_t)ORDER) || (stage_flags & 1u);      // fewer samples than taps, or no LMS stage wanted: everything passes through
  const uint32_t nmax = umax_wave(n);

  // the lane's run of block s0: samples [s0 + L SPL, + SPL), behind the long-term stage   src/SLAPredictor.c:1063-1099
  auto fetch_run = [&](uint32_t s0, int32_t (&v)[SPL]) {
    const uint32_t s = s0 + L * SPL;
    if (s + SPL <= n && (!use_ltm || s >= delay)) {
      tk_load_run<SPL>(in + s, v);
      if (use_ltm) {
        int32_t win[SPL + 4];                              // in[s - delay .. s - delay + SPL + ntaps - 2]
        const int32_t* w = in + (s - delay);
#pragma unroll
        for (int i = 0; i < SPL + 4; i++) { win[i] = ((uint32_t)i < SPL + ntaps - 1) ? w[i] : 0; }
#pragma unroll
        for (int i = 0; i < SPL; i++) {
          int64_t acc = (int64_t)1 << 30;
#pragma unroll
          for (int k = 0; k < 5; k++) { if ((uint32_t)k < ntaps) { acc += (int64_t)job.ltm_coef[k] * (int64_t)win[i + k]; } }
          v[i] = (int32_t)((uint32_t)v[i] - (uint32_t)(int32_t)(acc >> 31));
        }
      }
    } else {
#pragma unroll
      for (int i = 0; i < SPL; i++) {
        const uint32_t si = s + (uint32_t)i;
        int32_t x = 0;
        if (si < n) {
          x = in[si];
          if (use_ltm && si >= delay) {
            int64_t acc = (int64_t)1 << 30;
            for (uint32_t k = 0; k < ntaps; k++) { acc += (int64_t)job.ltm_coef[k] * (int64_t)in[si - delay + k]; }
            x = (int32_t)((uint32_t)x - (uint32_t)(int32_t)(acc >> 31));
          }
        }
        v[i] = x;
      }
    }
  };

  int addr[LPJ];                                     // ds_bpermute byte address of the job's lane l
#pragma unroll
  for (int l = 0; l < LPJ; l++) { addr[l] = (int)((row * 16 + (uint32_t)l * JPR + jr) * 4); }
  int32_t ca[K], cb[K], ha[K], hb[K], sa[K], sb[K];
#pragma unroll
  for (int a = 0; a < K; a++) { ca[a] = cb[a] = ha[a] = hb[a] = sa[a] = sb[a] = 0; }
  uint64_t fsum = 0;
  int32_t vm_next[SPL];
  fetch_run(0, vm_next);
  for (uint32_t s0 = 0; s0 < nmax; s0 += TAILK_BLK) {
    int32_t vm[SPL], em[SPL];
#pragma unroll
    for (int m = 0; m < SPL; m++) { vm[m] = vm_next[m]; em[m] = 0; }
    fetch_run(s0 + TAILK_BLK, vm_next);                // the next block travels while this one computes
    if (s0 == 0) { tailk_block<ORDER, K, true>(vm, em, addr, L, ca, cb, ha, hb, sa, sb); }
    else { tailk_block<ORDER, K, false>(vm, em, addr, L, ca, cb, ha, hb, sa, sb); }
    const uint32_t s = s0 + L * SPL;
#pragma unroll
    for (int m = 0; m < SPL; m++) {
      em[m] = pass ? vm[m] : em[m];
      if (s + (uint32_t)m < n) { fsum += (em[m] < 0) ? ~((uint32_t)em[m] << 1) : ((uint32_t)em[m] << 1); }      // zig-zag fold, src/SLAUtility.h:37
    }
    if (s + SPL <= n) { tk_store_run<SPL>(out + s, em); }
    else {
#pragma unroll
      for (int m = 0; m < SPL; m++) { if (s + (uint32_t)m < n) { out[s + m] = em[m]; } }
    }
  }
#pragma unroll
  for (int off = JPR; off < 16; off <<= 1) {
    const uint32_t lo = (uint32_t)__shfl_xor((int)(uint32_t)fsum, off);
    const uint32_t hi = (uint32_t)__shfl_xor((int)(uint32_t)(fsum >> 32), off);
    fsum += ((uint64_t)hi << 32) | lo;
  }
  if (have && L == 0) { fold_sum[j] = fsum; }
  span_end(span);
}


// ---------------------------------------------------------------------------------------------
// k_ltm_acf: autocorrelation of the lattice residual by the reference's real FFT
// (src/SLAPredictor.c:827-853, src/SLAUtility.c:220-312): forward real FFT -> |.|^2 -> inverse.
// One workgroup per (block, channel); the F doubles live in LDS (F <= 16384) or, for the largest
// encoder capacity, in an L2-resident global scratch slot.  Bit-exactness: twiddles are NOT
// evaluated here -- the host generates them once with the reference's sin()-seeded recurrence --
// and every butterfly is the same mul/mul/sub, mul/mul/add, sub, sub, add, add sequence; the order
// of butterflies inside a stage is free because they touch disjoint elements.
// Twiddle buffer layout (doubles): [0,F/2) stage re fwd | [F/2,F) stage im fwd | [F,3F/2) re inv |
// [3F/2,2F) im inv | [2F,2F+F/4) real-pass re fwd | +F/4 im fwd | +F/4 re inv | +F/4 im inv.
// ---------------------------------------------------------------------------------------------
#define ACF_THREADS 512

// LDS layout: complex slot c (re, im = 16 B, always moved as one b128 access) lives at slot c ^ S(c), S a GF(2)-linear map
// of the bits 3.. of c onto the low nibble.  Every access pattern of a power-of-two FFT is "base + j * 2^s": without S all
// lanes of an access group hit the same 4 banks (measured in round 1: 13 bank-conflict cycles per LDS instruction).
// Round 3 folded the higher nibbles onto the low one, c ^ ((c >> 4 ^ c >> 8 ^ c >> 12) & 15): 2.0 - 2.2 conflict cycles
// per LDS instruction were left (profiles/r3_sq_counters_*), and tests/tools/fft_lds_conflicts.py -- a model of every pass
// of k_ltm_acf2 on gfx950's real access groups: a ds_read_b128 is served 16 lanes at a time, but the lanes are
// {0-3, 12-15, 20-27}, {4-11, 16-19, 28-31} (+32), on 64 banks; a ds_write_b128 8 consecutive lanes at a time on 32 banks --
// reproduces those numbers (2.17 at 4096 points, 1.95 at 8192) and says where they come from: every WRITE of the first pass
// and of the inverse transform's first pass, every read at half-span 8, the bit-reversed scatter: all two-way.  The fold
// cannot separate what differs in bit 3 of c on a write (8 slots = 3 bits).  The rows below (bit j of the nibble = parity of
// (c >> 3) & ACF_SW_ROW<j>; hill-climbed in that script over the three transform sizes) leave 0.24 / 0.36 / 0.37 per
// instruction (2048 / 4096 / 8192 points), all of it in the middle pass's descending run.  The map costs about twice the
// instructions of the fold, and the kernel is bound by instruction issue, not by the LDS (the first build with these rows ran
// 7 % SLOWER: profiles/r4_fft_swizzle_ab.txt): it pays only together with the middle pass computing it four times per thread
// instead of four times per pair (acf2_middle).  S only reads bits >= 3 (bit 3 only into bits 0..2), so the map is a bijection, an index
// below 8 is its own image, and S(a + b) = S(a) ^ S(b) when a and b share no bit -- what the passes use.
#define ACF_SW_ROW0 718u
#define ACF_SW_ROW1 485u
#define ACF_SW_ROW2 129u
#define ACF_SW_ROW3 76u
static_assert((ACF_SW_ROW3 & 1u) == 0u, "bit 3 of the slot must not depend on itself");
__device__ __forceinline__ constexpr uint32_t acf_sw(uint32_t c)
{
  const uint32_t hi = c >> 3;
  return c ^ ((uint32_t)__builtin_popcount(hi & ACF_SW_ROW0) & 1u) ^ (((uint32_t)__builtin_popcount(hi & ACF_SW_ROW1) & 1u) << 1)
           ^ (((uint32_t)__builtin_popcount(hi & ACF_SW_ROW2) & 1u) << 2) ^ (((uint32_t)__builtin_popcount(hi & ACF_SW_ROW3) & 1u) << 3);
}

// One radix-2 butterfly of the reference's four1 loop (src/SLAUtility.c:220-260): same products, same order.
__device__ __forceinline__ void acf_bfly(double2& zi, double2& zq, double wr, double wi)
{
  const double tr = wr * zq.x - wi * zq.y;
  const double ti = wr * zq.y + wi * zq.x;
  zq = make_double2(zi.x - tr, zi.y - ti);
  zi = make_double2(zi.x + tr, zi.y + ti);
}

// R consecutive radix-2 stages (half-spans h, 2h, .., h<<(R-1)) on the 2^R points ci + m*h that only
// exchange data among themselves: one LDS round trip and one barrier instead of R, 2^R - 1 twiddles instead
// of R * 2^(R-1).  Every butterfly is the radix-2 one above on the same operands, so the results are the
// radix-2 results bit for bit (the order of butterflies inside a stage is free).
template <int R>
__device__ __forceinline__ void acf_pass(double2* z, uint32_t npts, uint32_t log2h, const double* __restrict__ twr,
                                         const double* __restrict__ twi)
{
  constexpr uint32_t P = 1u << R;
  const uint32_t h = 1u << log2h;
  for (uint32_t b = threadIdx.x; b < (npts >> R); b += ACF_THREADS) {
    const uint32_t low = b & (h - 1), ci = low + ((b >> log2h) << (log2h + R));
    double2 v[P];
#pragma unroll
    for (uint32_t m = 0; m < P; m++) { v[m] = z[acf_sw(ci + m * h)]; }
#pragma unroll
    for (int st = 0; st < R; st++) {
      const uint32_t hs = h << st;                      // half-span of this stage; its twiddles start at hs - 1
#pragma unroll
      for (uint32_t m = 0; m < P; m++) {
        if ((m >> st) & 1u) { continue; }
        const uint32_t k = low + (m & ((1u << st) - 1u)) * h;
        acf_bfly(v[m], v[m + (1u << st)], twr[hs - 1 + k], twi[hs - 1 + k]);
      }
    }
#pragma unroll
    for (uint32_t m = 0; m < P; m++) { z[acf_sw(ci + m * h)] = v[m]; }
  }
  __syncthreads();
}

__device__ __forceinline__ void acf_stages(double2* z, uint32_t log2npts, const double* __restrict__ twr,
                                           const double* __restrict__ twi)
{
  const uint32_t npts = 1u << log2npts;
  uint32_t log2h = 0;
  while (log2h < log2npts) {
    const uint32_t left = log2npts - log2h;
    if (left >= 3 && left != 4) { acf_pass<3>(z, npts, log2h, twr, twi); log2h += 3; }
    else if (left >= 2) { acf_pass<2>(z, npts, log2h, twr, twi); log2h += 2; }
    else { acf_pass<1>(z, npts, log2h, twr, twi); log2h += 1; }
  }
}

// the h1/h2 recombination pass shared by the forward and the inverse real transform; complex slot i-1
// holds the reference's (data[i1], data[i2]), slot npts-(i-1) its (data[i3], data[i4])
__device__ __forceinline__ void acf_real_pass(double2* z, uint32_t npts, double c2, const double* __restrict__ rtr,
                                              const double* __restrict__ rti)
{
  const double c1 = 0.5;
  for (uint32_t i = 2 + threadIdx.x; i <= (npts >> 1); i += ACF_THREADS) {
    const uint32_t ca = i - 1, cb = npts - (i - 1);
    const double wr = rtr[i - 2], wi = rti[i - 2];
    const double2 A = z[acf_sw(ca)], B = z[acf_sw(cb)];
    const double h1r = c1 * (A.x + B.x);
    const double h1i = c1 * (A.y - B.y);
    const double h2r = -c2 * (A.y + B.y);
    const double h2i = c2 * (A.x - B.x);
    z[acf_sw(ca)] = make_double2(h1r + wr * h2r - wi * h2i, h1i + wr * h2i + wi * h2r);
    z[acf_sw(cb)] = make_double2(h1r - wr * h2r + wi * h2i, -h1i + wr * h2i + wi * h2r);
  }
}

__device__ __forceinline__ double acf_at(const double2* z, uint32_t j)      // real element j of the array
{
  const double2 v = z[acf_sw(j >> 1)];
  return (j & 1u) ? v.y : v.x;
}

// Pitch candidate from the autocorrelation (reference src/SLAPredictor.c:866-924): segments run from an
// upward zero crossing to the next downward one (inclusive, capped at lag 256; a search that reaches 256
// inspects lags 256 and 257), each segment contributes its largest strict positive local maximum, and the
// earliest of the globally largest wins.  The three per-lag predicates are evaluated by 320 threads and
// ballotted into bit masks; one lane then hops from crossing to crossing with find-first-set and only
// touches the values of local maxima.
#define ACF_PICK_LAGS 320

__device__ __forceinline__ uint32_t first_set_from(const unsigned long long* m, uint32_t pos, uint32_t limit)
{
  while (pos < limit) {
    const unsigned long long w = m[pos >> 6] >> (pos & 63);
    if (w != 0) {
      const uint32_t hit = pos + (uint32_t)__builtin_ctzll(w);
      return (hit < limit) ? hit : limit;
    }
    pos = (pos | 63u) + 1u;
  }
  return limit;
}

__device__ __forceinline__ void acf_pick(const double* v, const unsigned long long* up, const unsigned long long* down,
                                         const unsigned long long* lm, uint32_t& chosen, uint32_t& ncand)
{
  double top = 0.0;
  uint32_t i = 1;
  chosen = 0; ncand = 0;
  while (i < 256) {
    uint32_t start = first_set_from(up, i, 256), end, arg = 0;
    double val = 0.0;
    end = (start < 256) ? first_set_from(down, start + 1, 256) : 257;
    for (uint32_t j = first_set_from(lm, start, end + 1); j <= end; j = first_set_from(lm, j + 1, end + 1)) {
      if (v[j] > val) { arg = j; val = v[j]; }
    }
    if (arg != 0) { ncand++; if (val > top) { top = val; chosen = arg; } }
    i = end + 1;
  }
}

// what a job leaves behind: the compact record {code, chosen lag, acf[0..4], acf[chosen-2..chosen+2]} (the Toeplitz
// solve follows in k_ltm_solve), or the first `head` autocorrelation values.
//
// The pitch scan (src/SLAPredictor.c:866-924; acf_pick above is its serial form) as ONE WAVE's work instead of one
// lane's -- a lane hopping from crossing to crossing through LDS-resident masks held the whole workgroup for 25 us per
// job, a third of the kernel.  The scan is a two-state machine over the lags 1 .. 255: OUT -> IN at an upward zero
// crossing, IN -> OUT behind a downward one (segment = [start, end], both inclusive; an unterminated segment also takes
// lag 256, and a scan that runs out of upward crossings while OUT inspects lags 256 and 257), and what it delivers is
// the first lag among the largest strict positive local maxima inside segments (per segment the first largest, over
// the segments the first largest: the smallest lag that attains the overall maximum) and whether there was any.
// Lane l owns lags 4l .. 4l+3: it folds its four transitions into one map {OUT, IN} -> {OUT, IN}, a 6-step wave scan
// composes the maps of the lanes below it, and an arg-max over (value, -lag) picks the candidate.
template <int THREADS>
__device__ __forceinline__ void acf_emit(const double2* z, uint32_t job, double* __restrict__ out, uint32_t head,
                                         double* s_acf, unsigned long long (*s_mask)[ACF_PICK_LAGS / 64])
{
  if (head == SLA_HIP_ACF_RECORD) {
    __syncthreads();
    if (threadIdx.x < ACF_PICK_LAGS) {
      const uint32_t j = threadIdx.x;
      const double vc = acf_at(z, j), vm = (j >= 1) ? acf_at(z, j - 1) : 0.0, vp = acf_at(z, j + 1);
      s_acf[j] = vc;
      const unsigned long long bu = __ballot(j >= 1 && j < 256 && vm < 0.0 && vc > 0.0);
      const unsigned long long bd = __ballot(j >= 1 && j < 256 && vc > 0.0 && vp < 0.0);
      const unsigned long long bl = __ballot(j >= 1 && j <= 257 && vc > vm && vc > vp && vc > 0.0);
      if ((j & 63) == 0) { s_mask[0][j >> 6] = bu; s_mask[1][j >> 6] = bd; s_mask[2][j >> 6] = bl; }
    }
    __syncthreads();
    if (threadIdx.x < 64) {
      const uint32_t lane = threadIdx.x;
      const uint32_t w = lane >> 4, sh = (lane & 15u) * 4;                      // lags 4*lane .. 4*lane+3 = bits sh .. sh+3 of word w
      const uint32_t up = (uint32_t)(s_mask[0][w] >> sh) & 15u, dn = (uint32_t)(s_mask[1][w] >> sh) & 15u;
      const uint32_t lm = (uint32_t)(s_mask[2][w] >> sh) & 15u;
      const uint32_t lm_hi = (uint32_t)s_mask[2][4] & 3u;                       // local maxima at lags 256, 257
      // this lane's map: state behind its four lags when the state in front of them is OUT (bit 0) / IN (bit 1)
      uint32_t f = 0;
#pragma unroll
      for (uint32_t s0 = 0; s0 < 2; s0++) {
        uint32_t st = s0;
#pragma unroll
        for (uint32_t q = 0; q < 4; q++) { st = st ? (((dn >> q) & 1u) ^ 1u) : ((up >> q) & 1u); }
        f |= st << s0;
      }
      // inclusive scan of the maps over the lanes (g = everything below, then f): h(s) = f(g(s))
      uint32_t g = f;
#pragma unroll
      for (uint32_t d = 1; d < 64; d <<= 1) {
        const uint32_t below = (uint32_t)__shfl_up((int)g, d);
        if (lane >= d) { g = ((g >> (below & 1u)) & 1u) | (((g >> ((below >> 1) & 1u)) & 1u) << 1); }
      }
      const uint32_t incl = g & 1u;                                            // state behind this lane's lags (the machine starts OUT)
      uint32_t st = (uint32_t)__shfl_up((int)incl, 1);
      if (lane == 0) { st = 0; }
      const uint32_t s255 = (uint32_t)__shfl((int)incl, 63);
      // s254 = state behind lag 254 = the state in front of lane 63's last lag
      double best = 0.0;
      uint32_t arg = 0;
      uint32_t s_before_last = 0;
#pragma unroll
      for (uint32_t q = 0; q < 4; q++) {
        const uint32_t lag = lane * 4 + q;
        const uint32_t prev = st;
        if (q == 3) { s_before_last = prev; }
        st = prev ? (((dn >> q) & 1u) ^ 1u) : ((up >> q) & 1u);
        if ((prev | st) && ((lm >> q) & 1u)) {                                   // inside a segment (both ends inclusive), a local maximum
          const double v = s_acf[lag];
          if (v > best) { best = v; arg = lag; }
        }
      }
      const uint32_t s254 = (uint32_t)__shfl((int)s_before_last, 63);
      if (lane < 2) {
        const uint32_t lag = 256 + lane;
        const bool open = (s255 != 0) && lane == 0;                            // unterminated segment: [start, 256]
        const bool pseudo = (s255 == 0 && s254 == 0);                          // OUT with no upward crossing left: lags 256, 257
        if ((open || pseudo) && ((lm_hi >> lane) & 1u)) {
          const double v = s_acf[lag];
          if (v > best) { best = v; arg = lag; }
        }
      }
      // arg-max over the lanes: larger value, then smaller lag (every candidate is > 0; 0.0 = none)
#pragma unroll
      for (int off = 32; off > 0; off >>= 1) {
        const double ov = __shfl_xor(best, off);
        const uint32_t oa = (uint32_t)__shfl_xor((int)arg, off);
        if (ov > best || (ov == best && ov > 0.0 && oa < arg)) { best = ov; arg = oa; }
      }
      double* o = out + (uint64_t)job * SLA_HIP_ACF_RECORD;
      const bool live = fabs(s_acf[0]) > (double)FLT_MIN;
      const uint32_t chosen = live ? arg : 0u;
      if (lane == 0) { o[0] = !live ? 0.0 : ((best > 0.0) ? 1.0 : 2.0); o[1] = (double)chosen; }      // 0: silent block, 2: no pitch candidate
      if (lane < 5) { o[2 + lane] = s_acf[lane]; }
      if (lane >= 8 && lane < 13) { const uint32_t k = lane - 8; o[7 + k] = (chosen + k >= 2) ? s_acf[chosen + k - 2] : 0.0; }
    }
  } else {
    for (uint32_t t = threadIdx.x; t < head; t += THREADS) { out[(uint64_t)job * head + t] = acf_at(z, t); }
  }
}

template <bool IN_LDS>
__global__ __launch_bounds__(ACF_THREADS)
void k_ltm_acf(const int32_t* __restrict__ res, uint64_t stride, const sla_hip_acf_job* __restrict__ jobs,
               uint32_t njobs, uint32_t log2F, const double* __restrict__ tw, double* __restrict__ scratch,
               double* __restrict__ out, uint32_t head, unsigned long long* span)
{
  extern __shared__ double2 lds2[];
  span_begin(span);
  __shared__ double s_acf[ACF_PICK_LAGS];
  __shared__ unsigned long long s_mask[3][ACF_PICK_LAGS / 64];
  const uint32_t F = 1u << log2F, npts = F >> 1, log2npts = log2F - 1;
  double2* z = IN_LDS ? lds2 : reinterpret_cast<double2*>(scratch + (uint64_t)blockIdx.x * F);
  const double* twr_f = tw;            const double* twi_f = tw + (F >> 1);
  const double* twr_i = tw + F;        const double* twi_i = tw + F + (F >> 1);
  const double* rtr_f = tw + 2 * F;    const double* rti_f = rtr_f + (F >> 2);
  const double* rtr_i = rti_f + (F >> 2); const double* rti_i = rtr_i + (F >> 2);
  const double scale = 4.656612873077392578125e-10;   // 2^-31

  for (uint32_t job = blockIdx.x; job < njobs; job += gridDim.x) {
    const sla_hip_acf_job jb = jobs[job];
    const int32_t* src = res + (uint64_t)jb.channel * stride + jb.blk_off;
    const uint32_t n = jb.blk_len;
    // load, zero-padded, straight into bit-reversed complex order
    for (uint32_t k = threadIdx.x; k < npts; k += ACF_THREADS) {
      const uint32_t j = __brev(k) >> (32 - log2npts);
      z[acf_sw(j)] = make_double2((2 * k < n) ? (double)src[2 * k] * scale : 0.0,
                                  (2 * k + 1 < n) ? (double)src[2 * k + 1] * scale : 0.0);
    }
    __syncthreads();
    acf_stages(z, log2npts, twr_f, twi_f);
    acf_real_pass(z, npts, -0.5, rtr_f, rti_f);
    __syncthreads();
    if (threadIdx.x == 0) {
      const double2 v = z[0];
      const double s0 = v.x + v.y, s1 = v.x - v.y;
      z[0] = make_double2(s0 * s0, s1 * s1);           // DC and Nyquist power
    }
    for (uint32_t i = 1 + threadIdx.x; i < npts; i += ACF_THREADS) {
      const double2 v = z[acf_sw(i)];
      z[acf_sw(i)] = make_double2(v.x * v.x + v.y * v.y, 0.0);
    }
    __syncthreads();
    acf_real_pass(z, npts, 0.5, rtr_i, rti_i);
    __syncthreads();
    if (threadIdx.x == 0) {
      const double2 v = z[0];
      z[0] = make_double2(0.5 * (v.x + v.y), 0.5 * (v.x - v.y));
    }
    __syncthreads();
    // in-place bit reversal, then the inverse stages
    for (uint32_t k = threadIdx.x; k < npts; k += ACF_THREADS) {
      const uint32_t j = __brev(k) >> (32 - log2npts);
      if (j > k) {
        const double2 a = z[acf_sw(k)], b = z[acf_sw(j)];
        z[acf_sw(k)] = b; z[acf_sw(j)] = a;
      }
    }
    __syncthreads();
    acf_stages(z, log2npts, twr_i, twi_i);
    acf_emit<ACF_THREADS>(z, job, out, head, s_acf, s_mask);
    __syncthreads();
  }
  span_end(span);
}

// ---------------------------------------------------------------------------------------------
// k_ltm_acf2: the same transform pair with fewer trips through the LDS (every butterfly is still the reference's
// radix-2 one on the same operands: bit for bit the results of k_ltm_acf).  Per job and 2^L complex points:
//   * the first three forward stages take their operands straight from the residual plane: the point at bit-reversed
//     position 8b + m is sample pair t + rev3(m) * 2^(L-3) when b = rev(t), so consecutive lanes read consecutive
//     samples, the zero padding costs no loads, and the scatter pass of k_ltm_acf is gone;
//   * forward recombination, power spectrum, inverse recombination and the inverse transform's bit reversal are ONE
//     pass: every step keeps the pair (i, N - i) to itself, so a thread carries its pairs through all of them in
//     registers, a barrier separates all reads from all (bit-reversed) writes -- four passes of k_ltm_acf;
//   * the record only needs autocorrelation lags 0 .. 259 (pitch <= 257, five taps): the last inverse passes only
//     compute the groups, and store the points, those lags depend on (`need` complex slots).
// LDS round trips per job at L = 13: 4 + 1 + ~3.9 instead of 15.
// ---------------------------------------------------------------------------------------------
typedef int32_t i32x2_u __attribute__((ext_vector_type(2), aligned(4)));
typedef uint32_t u32x4_t __attribute__((ext_vector_type(4)));

// one twiddle = one 16-byte buffer load from the table of (re, im) pairs: byte offset = per-lane part (VGPR) + a part that
// is the same for every lane (SGPR / literal) -- no address arithmetic per load (the split tables cost two 8-byte loads with
// 64-bit address arithmetic each: as many instructions as the butterfly they feed)
__device__ __forceinline__ double2 acf2_tw(__amdgpu_buffer_rsrc_t rsrc, uint32_t lane_bytes, uint32_t uniform_bytes)
{
  const u32x4_t v = __builtin_amdgcn_raw_buffer_load_b128(rsrc, lane_bytes, uniform_bytes, 0);
  return make_double2(__hiloint2double((int)v.y, (int)v.x), __hiloint2double((int)v.w, (int)v.z));
}

#define ACF2_NEED 162u        // complex slots that hold lags 0 .. 323: what acf_emit reads for the compact record

template <int R, int L, int THREADS>
__device__ __forceinline__ void acf2_first_pass(double2* z, const int32_t* __restrict__ src, uint32_t n,
                                                const double* __restrict__ twr, const double* __restrict__ twi)
{
  constexpr uint32_t P = 1u << R, ngroups = 1u << (L - R);
  const double scale = 4.656612873077392578125e-10;   // 2^-31
  uint32_t tid = threadIdx.x;
  asm volatile("" : "+v"(tid));
#pragma unroll 1
  for (uint32_t it = 0; it < (ngroups + THREADS - 1) / THREADS; it++) {
    const uint32_t t = tid + it * THREADS;
    if (t >= ngroups) { break; }
    const uint32_t b = __brev(t) >> (32 - (L - R));
    // Position P*b + m holds complex sample c = t + rev(m) * ngroups.  An odd m has rev(m) >= P/2, i.e. c >= npts/2: beyond any
    // block (blk_len <= capacity = npts), always zero padding.  The first stage pairs (m, m + 1) under the twiddle (1, 0): with
    // zq = +0 the reference's products and sums give tr = ti = +0 and both results equal zi bit for bit (an input is
    // (double)int * 2^-31, never -0): the stage is a copy, and the odd positions are neither loaded nor multiplied.
    double2 v[P];
#pragma unroll
    for (uint32_t m = 0; m < P; m += 2) {
      const uint32_t c = t + (__brev(m) >> (32 - R)) * ngroups;          // complex sample index of position P*b + m
      if (2 * c + 1 < n) {
        const i32x2_u w = *(const i32x2_u*)(src + 2 * c);
        v[m] = make_double2((double)w.x * scale, (double)w.y * scale);
      } else {
        v[m] = make_double2((2 * c < n) ? (double)src[2 * c] * scale : 0.0, 0.0);
      }
      v[m + 1] = v[m];
    }
    if (n > (1u << L)) {
      // (a caller of the launcher with blocks longer than half the transform: the odd positions hold samples; wave-uniform)
#pragma unroll
      for (uint32_t m = 1; m < P; m += 2) {
        const uint32_t c = t + (__brev(m) >> (32 - R)) * ngroups;
        v[m] = make_double2((2 * c < n) ? (double)src[2 * c] * scale : 0.0, (2 * c + 1 < n) ? (double)src[2 * c + 1] * scale : 0.0);
        acf_bfly(v[m - 1], v[m], twr[0], twi[0]);
      }
    }
#pragma unroll
    for (int st = 1; st < R; st++) {
      constexpr uint32_t one = 1u;
      const uint32_t hs = one << st;
#pragma unroll
      for (uint32_t m = 0; m < P; m++) {
        if ((m >> st) & 1u) { continue; }
        const uint32_t k = m & ((1u << st) - 1u);
        acf_bfly(v[m], v[m + (1u << st)], twr[hs - 1 + k], twi[hs - 1 + k]);      // the same twiddle in every lane: scalar loads
      }
    }
    const uint32_t a0 = acf_sw(P * b);                                    // (m < 8 is its own swizzle)
#pragma unroll
    for (uint32_t m = 0; m < P; m++) { z[a0 ^ m] = v[m]; }
  }
  __syncthreads();
}

// R radix-2 stages (half-spans h = 2^LOG2H, 2h, ..) in one trip through the LDS, every number the compiler can know a
// template parameter: the trip counts are 1 - 4, so whatever is computed per pass at run time (swizzle constants, twiddle
// offsets, loop bounds) is not amortised -- in the run-time form three of four issued instructions were such overhead.
//   * acf_sw is linear over GF(2) and the bits of m * h are clear in ci: acf_sw(ci + m * h) = acf_sw(ci) ^ acf_sw(m * h),
//     the second factor a literal;
//   * PRUNE (the compact record only needs the slots [0, ACF2_NEED) of the inverse transform): after this pass the
//     remaining stages only combine points whose positions agree modulo H = h << R, so only the positions p with
//     (p mod H) < need matter: groups beyond them are skipped, points beyond them not stored.
template <int R, int LOG2H, int L, int THREADS, bool INV, bool PRUNE>
__device__ __forceinline__ void acf2_pass(double2* z, __amdgpu_buffer_rsrc_t tw2)
{
  constexpr uint32_t P = 1u << R, h = 1u << LOG2H, H = h << R, npts = 1u << L, groups = npts >> R;
  constexpr uint32_t tw_base = INV ? npts : 0u;                             // first pair of this direction's stage table
  constexpr bool prune = PRUNE && (H > ACF2_NEED);
  constexpr uint32_t live_groups = (prune && h >= ACF2_NEED) ? (groups / h) * ACF2_NEED : groups;   // (only counts the work)
  (void)live_groups;
  // (an opaque copy of the thread number per pass: otherwise the addresses of EVERY pass are computed in front of the first
  // one -- the passes of a 4096-point job are straight-line code -- and held in registers, or spilled, until they are used)
  uint32_t tid = threadIdx.x;
  asm volatile("" : "+v"(tid));
#pragma unroll 1
  for (uint32_t it = 0; it < (groups + THREADS - 1) / THREADS; it++) {
    // pruned passes with h >= need: renumber the groups so that the live ones (low < need) are dense over the threads
    uint32_t low, blk;
    if (prune && h >= ACF2_NEED) {
      constexpr uint32_t per = ACF2_NEED;                                   // live groups per block of h
      const uint32_t g = tid + it * THREADS;
      if (it * THREADS >= (groups / h) * per) { break; }
      if (g >= (groups / h) * per) { continue; }
      blk = g / per; low = g - blk * per;
    } else {
      const uint32_t b = tid + it * THREADS;
      if (groups % THREADS != 0 && b >= groups) { continue; }
      low = b & (h - 1); blk = b >> LOG2H;
    }
    const uint32_t ci = low + (blk << (LOG2H + R));
    const uint32_t a0 = acf_sw(ci);
    double2 v[P];
#pragma unroll
    for (uint32_t m = 0; m < P; m++) { v[m] = z[a0 ^ acf_sw(m << LOG2H)]; }
#pragma unroll
    for (int st = 0; st < R; st++) {
      const uint32_t hs = h << st;
#pragma unroll
      for (uint32_t m = 0; m < P; m++) {
        if ((m >> st) & 1u) { continue; }
        // twiddle hs - 1 + low + (m mod 2^st) * h of this direction's table
        const double2 w = acf2_tw(tw2, low << 4, (tw_base + hs - 1 + (m & ((1u << st) - 1u)) * h) << 4);
        acf_bfly(v[m], v[m + (1u << st)], w.x, w.y);
      }
    }
#pragma unroll
    for (uint32_t m = 0; m < P; m++) { if (!prune || low + m * h < ACF2_NEED) { z[a0 ^ acf_sw(m << LOG2H)] = v[m]; } }
  }
  __syncthreads();
}

// the stage schedule of one direction from half-span 2^LOG2H on: passes of three stages, the tail as 2 + 2 or 2
template <int LOG2H, int L, int THREADS, bool INV, bool PRUNE>
__device__ __forceinline__ void acf2_stages(double2* z, __amdgpu_buffer_rsrc_t tw2)
{
  constexpr int left = L - LOG2H;
  if constexpr (left >= 3 && left != 4) {
    acf2_pass<3, LOG2H, L, THREADS, INV, PRUNE>(z, tw2);
    acf2_stages<LOG2H + 3, L, THREADS, INV, PRUNE>(z, tw2);
  } else if constexpr (left >= 2) {
    acf2_pass<2, LOG2H, L, THREADS, INV, PRUNE>(z, tw2);
    acf2_stages<LOG2H + 2, L, THREADS, INV, PRUNE>(z, tw2);
  } else if constexpr (left == 1) {
    acf2_pass<1, LOG2H, L, THREADS, INV, PRUNE>(z, tw2);
  }
}

// the recombination of the reference's realft (src/SLAUtility.c:262-312) on the pair A = slot i-1, B = slot npts-(i-1):
// the expressions of acf_real_pass, operation for operation
__device__ __forceinline__ void acf2_recombine(double2& A, double2& B, double c2, double wr, double wi)
{
  const double c1 = 0.5;
  const double h1r = c1 * (A.x + B.x);
  const double h1i = c1 * (A.y - B.y);
  const double h2r = -c2 * (A.y + B.y);
  const double h2i = c2 * (A.x - B.x);
  A = make_double2(h1r + wr * h2r - wi * h2i, h1i + wr * h2i + wi * h2r);
  B = make_double2(h1r - wr * h2r + wi * h2i, -h1i + wr * h2i + wi * h2r);
}

template <int L, int THREADS>
__device__ __forceinline__ void acf2_middle(double2* z, __amdgpu_buffer_rsrc_t tw2)
{
  // Thread tid takes the pairs (A = slot c, B = slot npts - c), c = tid + k THREADS in 1 .. npts/2 - 1 (c = 0: the DC / Nyquist
  // slot and the middle slot no pair touches, thread 0).  The swizzle is linear over GF(2) and tid, k THREADS share no bit:
  //   slot c            = tid ^ (k THREADS)                       -> S(tid) ^ literal
  //   slot npts - c     = u ^ ((Q - 1 - k) THREADS), u = THREADS - tid, Q = npts / THREADS   (tid = 0: (Q - k) THREADS, u = 0)
  // and the bit-reversed places of the scatter likewise (bit reversal is linear too): four swizzles per thread at run time
  // instead of four per pair (round 3 took c = tid + 1 + k THREADS, whose carry defeats this; its run also started one slot
  // off the 16-slot grid: two-way conflicts on every read of A).
  constexpr uint32_t npts = 1u << L, half = npts >> 1, K = half / THREADS, Q = npts / THREADS;
  static_assert(half % THREADS == 0 && K >= 1, "pairs per thread");
  double2 A[K], B[K];
  uint32_t tid = threadIdx.x;                                               // (opaque per phase, as in acf2_pass)
  asm volatile("" : "+v"(tid));
  const uint32_t u = (THREADS - tid) & (THREADS - 1u);
  const bool t0 = (tid == 0u);
  const uint32_t sa = acf_sw(tid), sb = acf_sw(u);
#pragma unroll
  for (uint32_t k = 0; k < K; k++) {
    if (k > 0 || !t0) {
      A[k] = z[sa ^ acf_sw(k * THREADS)];
      B[k] = z[sb ^ (t0 ? acf_sw((Q - k) * THREADS) : acf_sw((Q - 1u - k) * THREADS))];
    }
  }
  double2 dc = make_double2(0.0, 0.0), mid = make_double2(0.0, 0.0);
  if (threadIdx.x == 0) { dc = z[0]; mid = z[acf_sw(half)]; }
#pragma unroll
  for (uint32_t k = 0; k < K; k++) {
    if (k > 0 || !t0) {
      // recombination twiddles: pairs [2 npts, 2 npts + npts/2) forward, the next npts/2 inverse; entry c - 1 of either
      const double2 wf = acf2_tw(tw2, tid << 4, (2u * npts + k * THREADS - 1u) << 4);
      const double2 wi = acf2_tw(tw2, tid << 4, (2u * npts + half + k * THREADS - 1u) << 4);
      acf2_recombine(A[k], B[k], -0.5, wf.x, wf.y);
      A[k] = make_double2(A[k].x * A[k].x + A[k].y * A[k].y, 0.0);       // power spectrum  src/SLAPredictor.c:844-851
      B[k] = make_double2(B[k].x * B[k].x + B[k].y * B[k].y, 0.0);
      acf2_recombine(A[k], B[k], 0.5, wi.x, wi.y);
    }
    __builtin_amdgcn_sched_barrier(0);          // one pair at a time: hoisting every pair's twiddles and temporaries costs spills
  }
  if (threadIdx.x == 0) {
    const double s0 = dc.x + dc.y, s1 = dc.x - dc.y;
    const double2 pw = make_double2(s0 * s0, s1 * s1);                   // DC and Nyquist power  :839-842
    dc = make_double2(0.5 * (pw.x + pw.y), 0.5 * (pw.x - pw.y));
    mid = make_double2(mid.x * mid.x + mid.y * mid.y, 0.0);              // the slot no pair touches
  }
  __syncthreads();                                                        // every read above is done: the slots may be overwritten
  asm volatile("" : "+v"(tid));
  // rev_L(tid ^ k THREADS) = rev_L(tid) ^ rev_L(k THREADS)
  const uint32_t ra = acf_sw(__brev(tid) >> (32 - L)), rb = acf_sw(__brev(u) >> (32 - L));
#pragma unroll
  for (uint32_t k = 0; k < K; k++) {
    if (k > 0 || !t0) {
      z[ra ^ acf_sw(__brev(k * THREADS) >> (32 - L))] = A[k];
      z[rb ^ (t0 ? acf_sw(__brev((Q - k) * THREADS) >> (32 - L)) : acf_sw(__brev((Q - 1u - k) * THREADS) >> (32 - L)))] = B[k];
    }
  }
  if (threadIdx.x == 0) { z[0] = dc; z[acf_sw(1)] = mid; }                // rev(0) = 0, rev(npts/2) = 1
  __syncthreads();
}

// <= 128 registers: four waves per SIMD (two workgroups of 512 threads on 64 KiB of LDS each, or one of 1024 on 128 KiB)
template <int L, int ACF2_THREADS, bool RECORD>
__global__ __launch_bounds__(ACF2_THREADS, 4)
void k_ltm_acf2(const int32_t* __restrict__ res, uint64_t stride, const sla_hip_acf_job* __restrict__ jobs,
                uint32_t njobs, const double* __restrict__ tw, double* __restrict__ out, uint32_t head, unsigned long long* span)
{
  extern __shared__ double2 lds2[];
  span_begin(span);
  __shared__ double s_acf[ACF_PICK_LAGS];
  __shared__ unsigned long long s_mask[3][ACF_PICK_LAGS / 64];
  constexpr uint32_t npts = 1u << L, F = npts << 1;
  double2* z = lds2;
  const double* twr_f = tw;            const double* twi_f = tw + (F >> 1);      // split tables: the first pass (same twiddles in every lane)
  // the table of (re, im) pairs behind the split ones: [0, npts) forward stages | [npts, 2 npts) inverse | recombination
  const __amdgpu_buffer_rsrc_t tw2 = __builtin_amdgcn_make_buffer_rsrc((void*)(tw + 3 * (size_t)F), 0, (int)(3u * F * sizeof(double)), 0x00020000);
  for (uint32_t job = blockIdx.x; job < njobs; job += gridDim.x) {
    const sla_hip_acf_job jb = jobs[job];
    const int32_t* src = res + (uint64_t)jb.channel * stride + jb.blk_off;
    acf2_first_pass<3, L, ACF2_THREADS>(z, src, jb.blk_len, twr_f, twi_f);
    acf2_stages<3, L, ACF2_THREADS, false, false>(z, tw2);
    acf2_middle<L, ACF2_THREADS>(z, tw2);
    acf2_stages<0, L, ACF2_THREADS, true, RECORD>(z, tw2);
    acf_emit<ACF2_THREADS>(z, job, out, head, s_acf, s_mask);
    __syncthreads();
  }
  span_end(span);
}

// ---------------------------------------------------------------------------------------------
// k_ltm_solve: pitch and taps of every (block, channel) from k_ltm_acf's compact record, written straight into the
// job table k_tail reads -- the long-term stage never leaves the device (src/SLAPredictor.c:855-863, 913-979;
// the 5x5 LU solve with two refinement passes is src/SLAUtility.c:487-674; the tap quantiser src/SLAEncoder.c:629-640).
// One lane per job.  The reference accumulates the refinement residual in x87 long double: ext80 below is that
// arithmetic in integers (64-bit significand, round to nearest even), so the result is the reference's bit for bit.
// ---------------------------------------------------------------------------------------------
struct ext80 { uint64_t m; int32_t e; uint32_t neg; };      // (-1)^neg * m * 2^e, bit 63 of m set (m == 0: zero)

__device__ __forceinline__ ext80 ext_from_double(double d)
{
  ext80 r; r.m = 0; r.e = 0; r.neg = 0;
  const uint64_t bits = (uint64_t)__double_as_longlong(d);
  const uint32_t be = (uint32_t)(bits >> 52) & 0x7FFu;
  uint64_t frac = bits & 0xFFFFFFFFFFFFFull;
  r.neg = (uint32_t)(bits >> 63);
  if (be == 0) {
    if (frac == 0) { return r; }
    const int lz = __clzll((long long)frac);
    r.m = frac << lz; r.e = -1074 - lz;
    return r;
  }
  r.m = (frac | (1ull << 52)) << 11;
  r.e = (int32_t)be - 1075 - 11;
  return r;
}

// a + b rounded to a 64-bit significand (FADD with the x87 precision control at its default, extended)
__device__ __forceinline__ ext80 ext_add(ext80 a, ext80 b)
{
  if (b.m == 0) { return a; }
  if (a.m == 0) { return b; }
  if (b.e > a.e || (b.e == a.e && b.m > a.m)) { const ext80 t = a; a = b; b = t; }      // |a| >= |b|
  const uint32_t shift = (uint32_t)(a.e - b.e);
  const unsigned __int128 A = (unsigned __int128)a.m << 63;                                 // value = A * 2^(a.e - 63)
  unsigned __int128 B = 0;
  bool sticky = false;
  if (shift < 128) {
    const unsigned __int128 full = (unsigned __int128)b.m << 63;
    B = full >> shift;
    sticky = ((B << shift) != full);
  } else {
    sticky = true;
  }
  unsigned __int128 S;
  if (a.neg == b.neg) { S = A + B; } else { S = A - B - (sticky ? 1u : 0u); }            // true value = S + (0, 1) when sticky
  ext80 r; r.neg = a.neg; r.m = 0; r.e = 0;
  if (S == 0 && !sticky) { r.neg = 0; return r; }
  const uint64_t hi0 = (uint64_t)(S >> 64), lo0 = (uint64_t)S;
  const int lz = hi0 ? __clzll((long long)hi0) : 64 + __clzll((long long)lo0);
  S <<= lz;
  uint64_t m = (uint64_t)(S >> 64);
  const uint64_t low = (uint64_t)S;
  int32_t ex = a.e - 63 - lz + 64;
  const uint64_t half = 1ull << 63;
  const bool up = (low > half) || (low == half && (sticky || (m & 1ull)));
  if (up) { m += 1; if (m == 0) { m = half; ex += 1; } }
  r.m = m; r.e = ex;
  return r;
}

// (double)x: round the 64-bit significand to 53 bits, nearest even
__device__ __forceinline__ double ext_to_double(ext80 x)
{
  if (x.m == 0) { return x.neg ? -0.0 : 0.0; }
  uint64_t m53 = x.m >> 11;
  const uint64_t rem = x.m & 0x7FFull;
  if (rem > 0x400ull || (rem == 0x400ull && (m53 & 1ull))) { m53 += 1; }
  const double v = ldexp((double)m53, x.e + 11);           // m53 <= 2^53: exact
  return x.neg ? -v : v;
}

#define LTM_NT 5

// The solve with the tap count as a template parameter: every loop is unrolled and every index static, so the 5 x 5 work
// arrays live in registers (with run-time dimensions and the pivot's row index they sat in scratch memory: 66 us per
// C5 launch for 45 000 tiny solves).  The pivot row and the permuted right-hand side are picked by compare-and-select
// over the (at most five) candidates; the arithmetic and its order are the host's (sla_ltm.c), operation for operation.
template <int D>
__device__ __forceinline__ int ltm_lu_factor_s(double (&A)[D][D], uint32_t (&perm)[D], double (&scale)[D])
{
#pragma unroll
  for (int row = 0; row < D; row++) {
    double big = 0.0;
#pragma unroll
    for (int col = 0; col < D; col++) { if (fabs(A[row][col]) > big) { big = fabs(A[row][col]); } }
    if (fabs(big) <= (double)FLT_EPSILON) { return -1; }
    scale[row] = 1.0 / big;
  }
  int bad = 0;
#pragma unroll
  for (int col = 0; col < D; col++) {
#pragma unroll
    for (int row = 0; row < col; row++) {
      double sum = A[row][col];
#pragma unroll
      for (int k = 0; k < row; k++) { sum -= A[row][k] * A[k][col]; }
      A[row][col] = sum;
    }
    double big = 0.0;
    uint32_t imax = (uint32_t)col;
#pragma unroll
    for (int row = col; row < D; row++) {
      double sum = A[row][col];
#pragma unroll
      for (int k = 0; k < col; k++) { sum -= A[row][k] * A[k][col]; }
      A[row][col] = sum;
      const double t = scale[row] * fabs(sum);
      if (t >= big) { big = t; imax = (uint32_t)row; }
    }
#pragma unroll
    for (int r = col + 1; r < D; r++) {
      if (imax == (uint32_t)r) {
#pragma unroll
        for (int k = 0; k < D; k++) { const double t = A[r][k]; A[r][k] = A[col][k]; A[col][k] = t; }
        scale[r] = scale[col];
      }
    }
    perm[col] = imax;
    if (fabs(A[col][col]) <= (double)FLT_EPSILON) { bad = 1; }
    if (!bad && col != D - 1) {
      const double inv = 1.0 / A[col][col];
#pragma unroll
      for (int row = col + 1; row < D; row++) { A[row][col] *= inv; }
    }
  }
  return bad ? -1 : 0;
}

template <int D>
__device__ __forceinline__ void ltm_lu_substitute_s(const double (&A)[D][D], double (&b)[D], const uint32_t (&perm)[D])
{
  uint32_t first_nz = 0;
#pragma unroll
  for (int row = 0; row < D; row++) {
    const uint32_t pv = perm[row];
    double sum = b[row];
#pragma unroll
    for (int r = 0; r < D; r++) { if (pv == (uint32_t)r) { sum = b[r]; } }
    const double mine = b[row];
#pragma unroll
    for (int r = 0; r < D; r++) { if (pv == (uint32_t)r) { b[r] = mine; } }
    if (first_nz != 0) {
#pragma unroll
      for (int col = 0; col < row; col++) { if ((uint32_t)col >= first_nz) { sum -= A[row][col] * b[col]; } }
    } else if (sum != 0.0) {
      first_nz = (uint32_t)row;
    }
    b[row] = sum;
  }
#pragma unroll
  for (int row = D - 1; row >= 0; row--) {
    double sum = b[row];
#pragma unroll
    for (int col = row + 1; col < D; col++) { sum -= A[row][col] * b[col]; }
    b[row] = sum / A[row][row];
  }
}

template <int D>
__global__ __launch_bounds__(64)
void k_ltm_solve(const double* __restrict__ acf, const sla_hip_lpc_group* __restrict__ groups, uint32_t num_jobs,
                 sla_hip_tail_job* __restrict__ jobs)
{
  const uint32_t j = blockIdx.x * 64 + threadIdx.x;
  if (j >= num_jobs) { return; }
  const double* rec = acf + (uint64_t)j * SLA_HIP_ACF_RECORD;
  double low[5], mid[5];
#pragma unroll
  for (int i = 0; i < 5; i++) { low[i] = rec[2 + i]; mid[i] = rec[7 + i]; }
  const uint32_t chosen = (uint32_t)rec[1];
  double vec[LTM_NT] = {0.0, 0.0, 0.0, 0.0, 0.0};
  uint32_t pitch = 0;
  int ret = 0;
  if (rec[0] == 0.0) {
    ret = 0;                                               // silent residual: no pitch, zero taps
  } else if (rec[0] != 1.0 || chosen < (uint32_t)(D / 2 + 1)) {
    ret = 4;
  } else {
    double R[D][D], A[D][D], x[D], err[D], scale[D], b[D];
    uint32_t perm[D];
#pragma unroll
    for (int r = 0; r < D; r++) {
#pragma unroll
      for (int c = 0; c < D; c++) { R[r][c] = low[(r >= c) ? (r - c) : (c - r)]; A[r][c] = R[r][c]; }
    }
#pragma unroll
    for (int r = 0; r < D; r++) { b[r] = mid[2 + r - D / 2]; x[r] = b[r]; }
    if (ltm_lu_factor_s<D>(A, perm, scale) != 0) {
      ret = 4;
    } else {
      ltm_lu_substitute_s<D>(A, x, perm);
#pragma unroll 1
      for (int it = 0; it < 2; it++) {
#pragma unroll
        for (int r = 0; r < D; r++) {
          ext80 acc = ext_from_double(-b[r]);
#pragma unroll
          for (int c = 0; c < D; c++) { acc = ext_add(acc, ext_from_double(R[r][c] * x[c])); }
          err[r] = ext_to_double(acc);
        }
        ltm_lu_substitute_s<D>(A, err, perm);
#pragma unroll
        for (int r = 0; r < D; r++) { x[r] -= err[r]; }
      }
      double mag = 0.0;
#pragma unroll
      for (int r = 0; r < D; r++) { mag += fabs(x[r]); }
      if (mag >= 1.0) {
#pragma unroll
        for (int r = 0; r < D; r++) { x[r] = 0.0; }
        x[D / 2] = mid[2] / low[0];
      }
      pitch = chosen;
#pragma unroll
      for (int r = 0; r < D; r++) { vec[r] = x[r]; }
    }
  }
  if (ret != 0 || pitch >= 256u) { pitch = 0; }            // src/SLAEncoder.c:629-632
  const sla_hip_lpc_group g = groups[j];
  sla_hip_tail_job out;
  out.blk_off = g.pcm_off; out.blk_len = g.num_samples; out.channel = g.channel; out.pitch = pitch;
#pragma unroll
  for (int t = 0; t < LTM_NT; t++) {
    // Round(coef * 2^15) << 16 with the x86 conversion (out of range / NaN -> INT32_MIN)   src/SLAEncoder.c:635-640
    const double v = ((t < D) ? vec[t] : 0.0) * 32768.0;
    const double rv = (v >= 0.0) ? floor(v + 0.5) : -floor(-v + 0.5);
    const int32_t q = (!(rv > -2147483649.0 && rv < 2147483648.0)) ? INT32_MIN : (int32_t)rv;
    out.ltm_coef[t] = (int32_t)((uint32_t)q << 16);
  }
  out.pad_[0] = out.pad_[1] = 0;
  jobs[j] = out;
}

// ---------------------------------------------------------------------------------------------
// launchers (C-ABI, see include/sla_hip.h)
// ---------------------------------------------------------------------------------------------
// Optional extras of a launch travel in sla_hip_launch_extra (include/sla_hip.h): the two-word slot for its on-device execution
// span (see span_begin), a device-side group count, words to clear on the way.  (Rounds 2 - 3 passed them through thread-local
// one-shot requests that a later commit had to drop at every API entry; now they are arguments.)
static inline unsigned long long* span_of(const sla_hip_launch_extra* x) { return (x != nullptr) ? x->d_span : nullptr; }

// Tuning knobs of the launchers (include/sla_hip.h: sla_hip_tuning).  They belong to an encoder handle, which reads
// them ONCE (environment at SLAEncoder_Create, sla_hip_encoder_set_option afterwards) and names its copy to the
// launchers of its host thread; nothing on the launch path reads the environment.
// The thread keeps a COPY: a handle may be destroyed (or used on another thread) while this thread goes on calling
// launchers directly, and a stale pointer would then read freed memory.
static thread_local sla_hip_tuning t_tuning_val;
static thread_local bool t_tuning_set = false;
extern "C" void sla_hip_use_tuning(const sla_hip_tuning* tuning)
{
  if (tuning != nullptr) { t_tuning_val = *tuning; t_tuning_set = true; } else { t_tuning_set = false; }
}
static inline sla_hip_tuning tuning()
{
  sla_hip_tuning t;
  memset(&t, 0, sizeof(t));
  if (t_tuning_set) { t = t_tuning_val; }
  if (!(t.plan_margin >= PLAN_MARGIN)) { t.plan_margin = 0.0; }      // below the built-in margin: not a valid setting
  return t;
}

// hipFuncSetAttribute costs a driver call; the limit only ever has to grow (per kernel and device)
static hipError_t ensure_dynamic_lds(const void* fn, size_t bytes)
{
  static struct { const void* fn; int dev; size_t bytes; } seen[64];
  static int nseen = 0;
  static pthread_mutex_t mu = PTHREAD_MUTEX_INITIALIZER;
  int dev = 0;
  (void)hipGetDevice(&dev);
  pthread_mutex_lock(&mu);
  int slot = -1;
  for (int i = 0; i < nseen; i++) { if (seen[i].fn == fn && seen[i].dev == dev) { slot = i; break; } }
  if (slot >= 0 && seen[slot].bytes >= bytes) { pthread_mutex_unlock(&mu); return hipSuccess; }
  const hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
  if (e == hipSuccess) {
    if (slot < 0 && nseen < 64) { slot = nseen++; seen[slot].fn = fn; seen[slot].dev = dev; seen[slot].bytes = 0; }
    if (slot >= 0) { seen[slot].bytes = bytes; }
  }
  pthread_mutex_unlock(&mu);
  return e;
}

static inline int hip_rc(hipError_t e) { return (e == hipSuccess) ? 0 : -(int)e; }

#define LIST_LPC_GRID 256u        // list mode of k_lpc_blocks: workgroups that walk the list of uncertified blocks
#define LIST_FINISH_GRID 64u      // list mode of k_blocks_finish

static int launch_prepass_impl(const int32_t* d_pcm, uint64_t plane_stride, uint32_t num_channels,
                               uint32_t num_samples, uint32_t bits_per_sample, uint32_t mid_side,
                               uint32_t* d_or_mask, uint64_t* d_nz_mask, uint32_t* d_tile_or, sla_hip_stream_t stream)
{
  if (d_pcm == nullptr || d_or_mask == nullptr || d_nz_mask == nullptr) { return SLA_APIRESULT_INVALID_ARGUMENT; }
  if (num_channels == 0 || num_channels > 8 || bits_per_sample == 0 || bits_per_sample > 32
      || plane_stride < num_samples || (mid_side && num_channels != 2)) { return SLA_APIRESULT_INVALID_ARGUMENT; }
  hipStream_t st = (hipStream_t)stream;
  hipError_t e = hipMemsetAsync(d_or_mask, 0, 2 * sizeof(uint32_t), st);
  if (e != hipSuccess) { return hip_rc(e); }
  if (num_samples == 0) { return 0; }
  uint64_t nwords = ((uint64_t)num_samples + 63) / 64;
  uint32_t nblocks = (uint32_t)((nwords + 4 * PREPASS_WORDS - 1) / (4 * PREPASS_WORDS));
  const uint32_t shift = 32u - bits_per_sample;
  if (num_channels == 1) {
    hipLaunchKernelGGL(k_prepass<1>, dim3(nblocks), dim3(256), 0, st, d_pcm, plane_stride, num_channels, num_samples, shift, mid_side, d_or_mask, d_nz_mask, d_tile_or);
  } else if (num_channels == 2) {
    hipLaunchKernelGGL(k_prepass<2>, dim3(nblocks), dim3(256), 0, st, d_pcm, plane_stride, num_channels, num_samples, shift, mid_side, d_or_mask, d_nz_mask, d_tile_or);
  } else {
    hipLaunchKernelGGL(k_prepass<0>, dim3(nblocks), dim3(256), 0, st, d_pcm, plane_stride, num_channels, num_samples, shift, mid_side, d_or_mask, d_nz_mask, d_tile_or);
  }
  return hip_rc(hipGetLastError());
}

extern "C" int sla_hip_launch_prepass(const int32_t* d_pcm, uint64_t plane_stride, uint32_t num_channels,
                                      uint32_t num_samples, uint32_t bits_per_sample, uint32_t mid_side,
                                      uint32_t* d_or_mask, uint64_t* d_nz_mask, sla_hip_stream_t stream)
{
  return launch_prepass_impl(d_pcm, plane_stride, num_channels, num_samples, bits_per_sample, mid_side, d_or_mask, d_nz_mask, nullptr, stream);
}

// ---------------------------------------------------------------------------------------------
// k_batch_scan: what the host needs to know about every file of a batch (files back to back on 1024-sample boundaries,
// sla_hip_analyze_batch_device) from the prepass results, so that the silence mask itself stays on the device: the OR of the
// file's tile words (-> its offset_lshift), the number of all-zero 64-sample mask words inside it, and whether its last
// super-frame -- the one case in which a block can be SILENT without such a word: fewer than 127 samples left,
// src/SLAEncoder.c:392-408 with the minimum block length shrunk to what is left -- is all zero.  One workgroup per file.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256)
void k_batch_scan(const uint64_t* __restrict__ nz, const uint32_t* __restrict__ tile_or, const uint32_t* __restrict__ file_start,
                  const uint32_t* __restrict__ file_len, uint32_t max_block, uint32_t* __restrict__ info)
{
  __shared__ uint32_t s_or[4], s_zero[4];
  const uint32_t f = blockIdx.x, start = file_start[f], len = file_len[f];
  const uint32_t w0 = start >> 6, nw = len >> 6;                                    // whole mask words of the file (it starts on a word)
  const uint32_t t0 = start / SLA_HIP_PREPASS_TILE, nt = (len + SLA_HIP_PREPASS_TILE - 1) / SLA_HIP_PREPASS_TILE;
  uint32_t orw = 0, zeros = 0;
  for (uint32_t i = threadIdx.x; i < nw; i += 256) { zeros += (nz[w0 + i] == 0ull) ? 1u : 0u; }
  for (uint32_t i = threadIdx.x; i < nt; i += 256) { orw |= tile_or[t0 + i]; }
  for (int off = 32; off > 0; off >>= 1) { orw |= (uint32_t)__shfl_xor((int)orw, off); zeros += (uint32_t)__shfl_xor((int)zeros, off); }
  if ((threadIdx.x & 63) == 0) { s_or[threadIdx.x >> 6] = orw; s_zero[threadIdx.x >> 6] = zeros; }
  __syncthreads();
  if (threadIdx.x == 0) {
    const uint32_t rem = (max_block != 0u) ? (len % max_block) : 0u;
    uint32_t tail = 0;
    if (rem >= 1u && rem < 127u) {
      tail = 1;
      for (uint32_t p = start + len - rem; p < start + len; p++) { if ((nz[p >> 6] >> (p & 63u)) & 1ull) { tail = 0; break; } }
    }
    info[3 * f] = s_or[0] | s_or[1] | s_or[2] | s_or[3];
    info[3 * f + 1] = s_zero[0] + s_zero[1] + s_zero[2] + s_zero[3];
    info[3 * f + 2] = tail;
  }
}

extern "C" int sla_hip_launch_batch_scan(const uint64_t* d_nz_mask, const uint32_t* d_tile_or, const uint32_t* d_file_start,
                                         const uint32_t* d_file_len, uint32_t num_files, uint32_t max_block_samples,
                                         uint32_t* d_info, sla_hip_stream_t stream)
{
  if (d_nz_mask == nullptr || d_tile_or == nullptr || d_file_start == nullptr || d_file_len == nullptr || d_info == nullptr) { return SLA_APIRESULT_INVALID_ARGUMENT; }
  if (num_files == 0) { return 0; }
  hipLaunchKernelGGL(k_batch_scan, dim3(num_files), dim3(256), 0, (hipStream_t)stream, d_nz_mask, d_tile_or, d_file_start, d_file_len, max_block_samples, d_info);
  return hip_rc(hipGetLastError());
}

extern "C" int sla_hip_launch_prepass_tiles(const int32_t* d_pcm, uint64_t plane_stride, uint32_t num_channels,
                                            uint32_t num_samples, uint32_t bits_per_sample, uint32_t mid_side,
                                            uint32_t* d_or_mask, uint64_t* d_nz_mask, uint32_t* d_tile_or, sla_hip_stream_t stream)
{
  if (d_tile_or == nullptr) { return SLA_APIRESULT_INVALID_ARGUMENT; }
  return launch_prepass_impl(d_pcm, plane_stride, num_channels, num_samples, bits_per_sample, mid_side, d_or_mask, d_nz_mask, d_tile_or, stream);
}

static int launch_lpc_impl(const int32_t* d_pcm, uint64_t plane_stride, uint32_t mid_side, uint32_t order,
                           const sla_hip_lpc_group* d_groups, uint32_t num_groups, uint32_t max_window,
                           uint32_t max_cands_per_group,
                           const sla_hip_lpc_cand* d_cands, const double* d_window_pool,
                           double* d_out, int32_t* d_code, int32_t* d_kint, uint32_t* d_rshift,
                           sla_hip_stream_t stream, uint32_t mode_flags, uint32_t* d_rerun_counter, int32_t* d_lat_residual,
                           const uint32_t* list = nullptr, const uint32_t* list_count = nullptr, uint32_t* d_cert_flag = nullptr,
                           uint32_t audit_bps = 0u, unsigned long long* span = nullptr);

extern "C" int sla_hip_launch_lpc_x(const int32_t* d_pcm, uint64_t plane_stride, uint32_t mid_side, uint32_t order,
                                  const sla_hip_lpc_group* d_groups, uint32_t num_groups, uint32_t max_window,
                                  uint32_t max_cands_per_group,
                                  const sla_hip_lpc_cand* d_cands, const double* d_window_pool,
                                  double* d_out, int32_t* d_code, int32_t* d_kint, uint32_t* d_rshift,
                                  sla_hip_stream_t stream, const sla_hip_launch_extra* extra)
{
  return launch_lpc_impl(d_pcm, plane_stride, mid_side, order, d_groups, num_groups, max_window, max_cands_per_group, d_cands,
                         d_window_pool, d_out, d_code, d_kint, d_rshift, stream, 0u, nullptr, nullptr, nullptr, nullptr, nullptr, 0u, span_of(extra));
}

extern "C" int sla_hip_launch_lpc(const int32_t* d_pcm, uint64_t plane_stride, uint32_t mid_side, uint32_t order,
                                  const sla_hip_lpc_group* d_groups, uint32_t num_groups, uint32_t max_window,
                                  uint32_t max_cands_per_group,
                                  const sla_hip_lpc_cand* d_cands, const double* d_window_pool,
                                  double* d_out, int32_t* d_code, int32_t* d_kint, uint32_t* d_rshift,
                                  sla_hip_stream_t stream)
{
  return sla_hip_launch_lpc_x(d_pcm, plane_stride, mid_side, order, d_groups, num_groups, max_window, max_cands_per_group, d_cands, d_window_pool, d_out, d_code, d_kint, d_rshift, stream, nullptr);
}

extern "C" int sla_hip_launch_lpc_blocks_x(const int32_t* d_pcm, uint64_t plane_stride, uint32_t mid_side, uint32_t order,
                                         const sla_hip_lpc_group* d_groups, uint32_t num_groups, uint32_t max_window,
                                         const sla_hip_lpc_cand* d_cands, const double* d_window_pool,
                                         double* d_out, int32_t* d_code, int32_t* d_kint, uint32_t* d_rshift,
                                         int32_t* d_lattice_residual, sla_hip_stream_t stream, const sla_hip_launch_extra* extra)
{
  if (d_code == nullptr || d_lattice_residual == nullptr) { return SLA_APIRESULT_INVALID_ARGUMENT; }
  return launch_lpc_impl(d_pcm, plane_stride, mid_side, order, d_groups, num_groups, max_window, 1, d_cands,
                         d_window_pool, d_out, d_code, d_kint, d_rshift, stream, 0u, nullptr, d_lattice_residual, nullptr, nullptr, nullptr, 0u, span_of(extra));
}

extern "C" int sla_hip_launch_lpc_blocks(const int32_t* d_pcm, uint64_t plane_stride, uint32_t mid_side, uint32_t order,
                                         const sla_hip_lpc_group* d_groups, uint32_t num_groups, uint32_t max_window,
                                         const sla_hip_lpc_cand* d_cands, const double* d_window_pool,
                                         double* d_out, int32_t* d_code, int32_t* d_kint, uint32_t* d_rshift,
                                         int32_t* d_lattice_residual, sla_hip_stream_t stream)
{
  return sla_hip_launch_lpc_blocks_x(d_pcm, plane_stride, mid_side, order, d_groups, num_groups, max_window, d_cands, d_window_pool, d_out, d_code, d_kint, d_rshift, d_lattice_residual, stream, nullptr);
}

extern "C" int sla_hip_launch_lpc_blocks_cert_x(const int32_t* d_pcm, uint64_t plane_stride, uint32_t mid_side, uint32_t order,
                                              const sla_hip_lpc_group* d_groups, uint32_t num_groups, uint32_t max_window,
                                              const sla_hip_lpc_cand* d_cands, const double* d_window_pool,
                                              double* d_out, int32_t* d_code, int32_t* d_kint, uint32_t* d_rshift,
                                              uint32_t* d_cert_flag, uint32_t* d_fallback_list, uint32_t* d_fallback_count,
                                              double safety, uint32_t bits_per_sample, sla_hip_stream_t stream,
                                                const sla_hip_launch_extra* extra)
{
  if (d_pcm == nullptr || d_groups == nullptr || d_cands == nullptr || d_window_pool == nullptr || d_out == nullptr || d_code == nullptr
      || d_kint == nullptr || d_rshift == nullptr || d_cert_flag == nullptr || d_fallback_list == nullptr || d_fallback_count == nullptr) {
    return SLA_APIRESULT_INVALID_ARGUMENT;
  }
  if (order < 1 || max_window == 0 || bits_per_sample == 0 || bits_per_sample > 32 || !(safety >= 1.0)) { return SLA_APIRESULT_INVALID_ARGUMENT; }
  const uint32_t lags = sla_hip_search_exact_lags(order);
  if (lags == 0) { return SLA_APIRESULT_EXCEED_HANDLE_CAPACITY; }      // orders above 52: the exact kernels only
  if (num_groups == 0) { return 0; }
  hipStream_t st = (hipStream_t)stream;
  hipError_t e = hipSuccess;
  unsigned long long* span = span_of(extra);
  const uint32_t* dyn = (extra != nullptr) ? extra->d_group_count : nullptr;      // num_groups is then an upper bound, the kernels read the number
  const dim3 grid((num_groups + 3) / 4), block(256);
#define SLA_ACFB(NBB) do { \
    if (mid_side) { hipLaunchKernelGGL((k_acf_blocks<NBB, true>), grid, block, 0, st, d_pcm, plane_stride, order, d_groups, num_groups, d_window_pool, d_out, d_rshift, span, d_fallback_count, dyn); } \
    else { hipLaunchKernelGGL((k_acf_blocks<NBB, false>), grid, block, 0, st, d_pcm, plane_stride, order, d_groups, num_groups, d_window_pool, d_out, d_rshift, span, d_fallback_count, dyn); } } while (0)
  switch (lags) {
    case 12: SLA_ACFB(3); break;
    case 20: SLA_ACFB(5); break;
    case 36: SLA_ACFB(9); break;
    default: SLA_ACFB(13); break;
  }
#undef SLA_ACFB
  e = hipGetLastError();
  if (e != hipSuccess) { return hip_rc(e); }
  const dim3 fgrid((num_groups + 63) / 64), fblock(64);
  const uint32_t audit_every = tuning().cert_audit;
#define SLA_FINC(PP) hipLaunchKernelGGL((k_blocks_finish<PP, true>), fgrid, fblock, 0, st, d_groups, num_groups, order, d_out, d_code, d_kint, \
                                        d_rshift, (const uint32_t*)nullptr, (const uint32_t*)nullptr, d_cert_flag, d_fallback_list, d_fallback_count, safety, bits_per_sample, dyn, audit_every)
  if (order <= 16) { SLA_FINC(16); } else if (order <= 32) { SLA_FINC(32); } else if (order <= 48) { SLA_FINC(48); } else { SLA_FINC(64); }
#undef SLA_FINC
  e = hipGetLastError();
  if (e != hipSuccess) { return hip_rc(e); }
  // whatever could not be certified: the exact kernels over the list the finish kernel left (usually empty)
  return launch_lpc_impl(d_pcm, plane_stride, mid_side, order, d_groups, num_groups, max_window, 1, d_cands,
                         d_window_pool, d_out, d_code, d_kint, d_rshift, stream, 0u, nullptr, nullptr,
                         d_fallback_list, d_fallback_count, d_cert_flag, bits_per_sample);
}

extern "C" int sla_hip_launch_lpc_blocks_cert(const int32_t* d_pcm, uint64_t plane_stride, uint32_t mid_side, uint32_t order,
                                              const sla_hip_lpc_group* d_groups, uint32_t num_groups, uint32_t max_window,
                                              const sla_hip_lpc_cand* d_cands, const double* d_window_pool,
                                              double* d_out, int32_t* d_code, int32_t* d_kint, uint32_t* d_rshift,
                                              uint32_t* d_cert_flag, uint32_t* d_fallback_list, uint32_t* d_fallback_count,
                                              double safety, uint32_t bits_per_sample, sla_hip_stream_t stream)
{
  return sla_hip_launch_lpc_blocks_cert_x(d_pcm, plane_stride, mid_side, order, d_groups, num_groups, max_window, d_cands, d_window_pool, d_out, d_code,
                                          d_kint, d_rshift, d_cert_flag, d_fallback_list, d_fallback_count, safety, bits_per_sample, stream, nullptr);
}

extern "C" int sla_hip_launch_lpc_rerun(const int32_t* d_pcm, uint64_t plane_stride, uint32_t mid_side, uint32_t order,
                                        const sla_hip_lpc_group* d_groups, uint32_t num_groups, uint32_t max_window,
                                        uint32_t max_cands_per_group, const sla_hip_lpc_cand* d_cands,
                                        double* d_out, uint32_t* d_rerun_counter, sla_hip_stream_t stream)
{
  return launch_lpc_impl(d_pcm, plane_stride, mid_side, order, d_groups, num_groups, max_window, max_cands_per_group, d_cands,
                         nullptr, d_out, nullptr, nullptr, nullptr, stream, 128u, d_rerun_counter, nullptr);
}

static int launch_lpc_impl(const int32_t* d_pcm, uint64_t plane_stride, uint32_t mid_side, uint32_t order,
                           const sla_hip_lpc_group* d_groups, uint32_t num_groups, uint32_t max_window,
                           uint32_t max_cands_per_group,
                           const sla_hip_lpc_cand* d_cands, const double* d_window_pool,
                           double* d_out, int32_t* d_code, int32_t* d_kint, uint32_t* d_rshift,
                           sla_hip_stream_t stream, uint32_t mode_flags, uint32_t* d_rerun_counter, int32_t* d_lat_residual,
                           const uint32_t* list, const uint32_t* list_count, uint32_t* d_cert_flag, uint32_t audit_bps,
                           unsigned long long* span)
{
  if (d_pcm == nullptr || d_groups == nullptr || d_cands == nullptr || d_out == nullptr) { return SLA_APIRESULT_INVALID_ARGUMENT; }
  if (order < 1 || order > 255 || max_window == 0 || max_cands_per_group == 0) { return SLA_APIRESULT_INVALID_ARGUMENT; }
  if ((d_code != nullptr) != (d_kint != nullptr) || (d_code != nullptr) != (d_rshift != nullptr)) { return SLA_APIRESULT_INVALID_ARGUMENT; }
  if (d_code != nullptr && max_cands_per_group != 1) { return SLA_APIRESULT_INVALID_ARGUMENT; }
  if (num_groups == 0) { return 0; }
  const sla_hip_tuning tune = tuning();
  if (list != nullptr && (d_code == nullptr || order > 64 || d_cert_flag == nullptr)) { return SLA_APIRESULT_INVALID_ARGUMENT; }
  if (d_code != nullptr && order <= 64 && (!tune.lpc_blocks_chains || list != nullptr)) {
    // chosen blocks: term tiles + lane-parallel Levinson (k_lpc_blocks)
    if (d_window_pool == nullptr) { return SLA_APIRESULT_INVALID_ARGUMENT; }
    uint32_t pmax = 64 / order;
    if (pmax > LPC_MAX_PACK) { pmax = LPC_MAX_PACK; }
    if (pmax > 2) { pmax = 2; }
    if (tune.lpc_pack >= 1 && tune.lpc_pack <= LPC_MAX_PACK && tune.lpc_pack * order <= 64) { pmax = tune.lpc_pack; }
    if (list != nullptr) { pmax = 1; }      // list mode (usually an empty list): the smallest LDS footprint, so that the launch does not wait for whole CUs
    for (uint32_t p = pmax; p >= 1; p--) {
      uint32_t nch = 16;
      while (nch < p * order) { nch <<= 1; }
      const size_t xr = ((size_t)max_window + 1) & ~(size_t)1;       // even: the term rows behind the windows are read as 16-byte pairs
      const uint32_t spl = (nch <= 32) ? 12 : 6;                    // producer lanes per chain: nch * spl <= LB_PRODUCERS
      // steps per tile: 48 for the wide packs (64 chains, 6 producer lanes each: C5-120 s 10.7 -> 10.3 ms per step, a
      // ten-minute order-16 file loses with them: 2.31 -> 2.46 ms) when the longer tiles still fit, else 24
      uint32_t lbk = (spl == 6 && 2u * 48u * p <= LB_PRODUCERS && tune.lpc_tile != 24u && list == nullptr) ? 48u : 24u;
      size_t bytes = sizeof(double) * ((size_t)p * xr + (size_t)2 * (lbk + 2) * nch + (size_t)2 * 2 * lbk * p + (size_t)p * (order + 1));
      if (bytes > SLA_HIP_LDS_BUDGET && lbk == 48u) {
        lbk = 24u;
        bytes = sizeof(double) * ((size_t)p * xr + (size_t)2 * (lbk + 2) * nch + (size_t)2 * 2 * lbk * p + (size_t)p * (order + 1));
      }
      if (bytes > SLA_HIP_LDS_BUDGET) { continue; }
      const void* fn = (spl == 12) ? (const void*)k_lpc_blocks<12, 24> : (lbk == 48u) ? (const void*)k_lpc_blocks<6, 48> : (const void*)k_lpc_blocks<6, 24>;
      hipError_t e = ensure_dynamic_lds(fn, bytes);
      if (e != hipSuccess) { return hip_rc(e); }
      // list mode: a fixed grid walks the list (two workgroups per CU hold what the device can run at once)
      const dim3 grid((list != nullptr) ? LIST_LPC_GRID : (num_groups + p - 1) / p), block(LB_THREADS);
#ifdef SLA_HIP_DEBUG
      const uint32_t clk = (uint32_t)(getenv("SLA_HIP_LPC_CLK") != nullptr);
#else
      const uint32_t clk = 0;
#endif
      // Levinson-Durbin + quantiser: in registers, one lane per window (k_blocks_finish), unless the lattice is fused in
      // (it needs the coefficients inside the workgroup)
      const uint32_t defer = (d_lat_residual == nullptr && order <= 64) ? 1u : 0u;
      if (spl == 12) {
        hipLaunchKernelGGL((k_lpc_blocks<12, 24>), grid, block, bytes, (hipStream_t)stream, d_pcm, plane_stride, mid_side, order,
                           d_groups, num_groups, p, d_window_pool, d_out, d_code, d_kint, d_rshift, (uint32_t)xr, nch, clk, span, d_lat_residual, defer, list, list_count);
      } else if (lbk == 48u) {
        hipLaunchKernelGGL((k_lpc_blocks<6, 48>), grid, block, bytes, (hipStream_t)stream, d_pcm, plane_stride, mid_side, order,
                           d_groups, num_groups, p, d_window_pool, d_out, d_code, d_kint, d_rshift, (uint32_t)xr, nch, clk, span, d_lat_residual, defer, list, list_count);
      } else {
        hipLaunchKernelGGL((k_lpc_blocks<6, 24>), grid, block, bytes, (hipStream_t)stream, d_pcm, plane_stride, mid_side, order,
                           d_groups, num_groups, p, d_window_pool, d_out, d_code, d_kint, d_rshift, (uint32_t)xr, nch, clk, span, d_lat_residual, defer, list, list_count);
      }
      if (defer) {
        const dim3 fgrid((list != nullptr) ? LIST_FINISH_GRID : (num_groups + 63) / 64), fblock(64);
#define SLA_FIN(PP) hipLaunchKernelGGL((k_blocks_finish<PP, false>), fgrid, fblock, 0, (hipStream_t)stream, d_groups, num_groups, order, d_out, d_code, d_kint, \
                                       d_rshift, list, list_count, d_cert_flag, (uint32_t*)nullptr, (uint32_t*)nullptr, 0.0, audit_bps)
        if (order <= 16) { SLA_FIN(16); } else if (order <= 32) { SLA_FIN(32); } else if (order <= 48) { SLA_FIN(48); } else { SLA_FIN(64); }
#undef SLA_FIN
      }
#ifdef SLA_HIP_DEBUG
      if (clk) {
        unsigned long long h[8] = {0}, z[8] = {0};
        (void)hipStreamSynchronize((hipStream_t)stream);
        (void)hipMemcpyFromSymbol(h, HIP_SYMBOL(g_lpc_clk), sizeof(h));
        (void)hipMemcpyToSymbol(HIP_SYMBOL(g_lpc_clk), z, sizeof(z));
        int occ = 0; (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, fn, LB_THREADS, bytes);
        if (h[3] != 0) {
          fprintf(stderr, "[k_lpc_blocks] LDS %zu B, %d workgroups per CU; ", bytes, occ);
          fprintf(stderr, "[k_lpc_blocks] %llu workgroups (pack %u), ticks per workgroup: stage %llu, chains %llu (lag consumer busy %llu, producer busy %llu, energy consumer busy %llu), levinson+quantiser %llu\n",
                  h[3], p, h[0] / h[3], h[1] / h[3], h[4] / h[3], h[5] / h[3], h[6] / h[3], h[2] / h[3]);
        }
      }
#endif
      return hip_rc(hipGetLastError());
    }
    return SLA_APIRESULT_EXCEED_HANDLE_CAPACITY;
  }
  if (d_lat_residual != nullptr) { return SLA_APIRESULT_INVALID_ARGUMENT; }      /* only k_lpc_blocks carries the lattice */
  // windows per workgroup: as many as the LDS budget takes, at most what fills the three chain waves
  size_t x_region = (size_t)max_window;
  const size_t per_group_r = (size_t)max_cands_per_group * (order + 1);
  uint32_t pack = 1;
  for (uint32_t p = 2 /* measured on C2/C3: 2 beats 1 and 4 */; p >= 1; p--) {
    size_t xr = (size_t)max_window;
    if (xr * p < 2 * (size_t)p * max_cands_per_group * (order + 2)) { xr = 2 * (size_t)max_cands_per_group * (order + 2); }
    const size_t bytes = sizeof(double) * ((size_t)p * xr + (size_t)p * per_group_r);
    if (bytes <= SLA_HIP_LDS_BUDGET && ((size_t)(p - 1) * max_cands_per_group * order < 192 || p == 1)) { pack = p; x_region = xr; break; }
  }
  if (tune.lpc_pack >= 1 && tune.lpc_pack < pack) { pack = tune.lpc_pack; }
  if (x_region * pack < 2 * (size_t)pack * max_cands_per_group * (order + 2)) { x_region = 2 * (size_t)max_cands_per_group * (order + 2); }
  size_t lds = sizeof(double) * ((size_t)pack * x_region + (size_t)pack * per_group_r);
  if (lds > SLA_HIP_LDS_BUDGET) { return SLA_APIRESULT_EXCEED_HANDLE_CAPACITY; }
  hipError_t e = ensure_dynamic_lds((const void*)k_lpc, lds);
  if (e != hipSuccess) { return hip_rc(e); }
  // a group with many chains (the partition search: candidates x lags) gets 7 chain waves instead of 3: the window
  // owns the LDS, so the workgroups per CU are few and more waves per workgroup are what hides the LDS gathers
  // (C5: 14.4 -> 13.3 ms of search per minute of audio)
  uint32_t lpc_threads = ((size_t)pack * max_cands_per_group * order >= 448) ? 512u : 256u;
  if (tune.lpc_threads == 256 || tune.lpc_threads == 512) { lpc_threads = tune.lpc_threads; }
  hipLaunchKernelGGL(k_lpc, dim3((num_groups + pack - 1) / pack), dim3(lpc_threads), lds, (hipStream_t)stream, d_pcm, plane_stride, mid_side, order,
                     d_groups, num_groups, pack, d_cands, d_window_pool, d_out, d_code, d_kint, d_rshift, (uint32_t)x_region,
                     mode_flags, d_rerun_counter);
  return hip_rc(hipGetLastError());
}

// ---------------------------------------------------------------------------------------------
// k_search_cert: the certificate of the windows over the exactness limit (k_search_finish's mode 1) with its lanes full.
// A candidate needs TWO Schur recursions, one per end of the bracket; k_search_finish ran them one after the other in the
// candidate's lane, and an 8192-sample window has 36 candidates: 36 of 64 lanes busy, twice.  Here a lane is (candidate, end):
// a wave takes 32 consecutive candidate SLOTS of the launch -- the slots of the search groups are numbered consecutively,
// so a slot number is a flat candidate index and a wave's slots belong to at most two groups of equal size (it finds the
// group from the quotient and walks on where groups are shorter) -- lanes 0..31 run the end r0 + d, lanes 32..63 the end
// r0 - d of the same 32 candidates, the results meet through one cross-half exchange.  Same values as mode 1 (the same
// schur_error on the same operands); 14 KB of sums per wave become 12.5.
// ---------------------------------------------------------------------------------------------
#define XC_CANDS 32
template <int P>
__global__ __launch_bounds__(64)
void k_search_cert(uint32_t order, uint32_t lags, uint32_t per_max,
                   const sla_hip_lpc_group* __restrict__ groups, uint32_t num_groups, const sla_hip_lpc_cand* __restrict__ cands,
                   const double* __restrict__ tile_sums, double* __restrict__ out, double exact_limit, double cert,
                   uint32_t* __restrict__ any_exact)
{
  extern __shared__ __attribute__((aligned(16))) double lds[];          // r[XC_CANDS][order + 1]
  __shared__ uint32_t s_grp[XC_CANDS], s_start[XC_CANDS], s_len[XC_CANDS], s_live[XC_CANDS];
  __shared__ double s_energy[XC_CANDS];
  const uint32_t O1 = order + 1, O2 = order + 2;
  const uint32_t lane = threadIdx.x, half = lane >> 5, cl = lane & 31u;
  const uint32_t slot0 = groups[0].slot_first;
  const uint32_t total = groups[num_groups - 1].slot_first + groups[num_groups - 1].cand_count - slot0;
  const uint32_t s = blockIdx.x * XC_CANDS + cl;                         // slot of this launch (both halves: the same candidate)
  if (blockIdx.x * XC_CANDS >= total) { return; }
  if (half == 0) {
    uint32_t live = 0, g = 0, start = 0, len = 0;
    double energy = 0.0;
    if (s < total) {
      // no group has more than per_max candidates: group index >= s / per_max; walk on over shorter groups
      g = s / per_max;
      if (g >= num_groups) { g = num_groups - 1; }
      for (int i = 0; i < 4 && g + 1 < num_groups && groups[g + 1].slot_first - slot0 <= s; i++) { g++; }
      if (g + 1 < num_groups && groups[g + 1].slot_first - slot0 <= s) {
        // many shorter groups in front (a batch of files, each ending in a short super-frame): bisect the rest
        uint32_t lo = g + 1, hi = num_groups - 1;
        while (lo < hi) {
          const uint32_t mid = (lo + hi + 1) >> 1;
          if (groups[mid].slot_first - slot0 <= s) { lo = mid; } else { hi = mid - 1; }
        }
        g = lo;
      }
      const sla_hip_lpc_group gr = groups[g];
      const uint32_t ci = s - (gr.slot_first - slot0);
      if (ci < gr.cand_count) {
        const uint32_t ntiles = (gr.num_samples + SLA_HIP_XTILE - 1) / SLA_HIP_XTILE;
        const double* ts = tile_sums + (uint64_t)g * SLA_HIP_XTILES * 2 * lags;
        for (uint32_t t = 0; t < ntiles; t++) { energy += ts[(uint64_t)t * 2 * lags]; }
        if (energy < exact_limit) {
          if (any_exact != nullptr) { atomicOr(any_exact, 1u); }        // an exact window: the mode-2 launch behind this one takes it
        } else {
          const sla_hip_lpc_cand cd = cands[gr.cand_first + ci];
          live = 1; start = cd.start; len = cd.len;
        }
      }
    }
    s_grp[cl] = g; s_start[cl] = start; s_len[cl] = len; s_live[cl] = live; s_energy[cl] = energy;
  }
  __syncthreads();
  double* r = lds;
  for (uint32_t q = lane; q < XC_CANDS * O1; q += 64) {
    const uint32_t c2 = q / O1, lag = q - c2 * O1;
    if (s_live[c2]) {
      const double* ts = tile_sums + (uint64_t)s_grp[c2] * SLA_HIP_XTILES * 2 * lags;
      const uint32_t st = s_start[c2], ln = s_len[c2], end = st + ln;
      double sum = 0.0;
      if (lag < ln) {
        const uint32_t tl = (end - 1) / SLA_HIP_XTILE;
        for (uint32_t t = st / SLA_HIP_XTILE; t <= tl; t++) { sum += ts[(uint64_t)t * 2 * lags + lag]; }
        sum -= ts[(uint64_t)tl * 2 * lags + lags + lag];
      }
      r[c2 * O1 + lag] = sum;
    }
  }
  __syncthreads();
  const double inf = __longlong_as_double(0x7FF0000000000000ll);
  const double nan = __longlong_as_double(0x7FF8000000000000ll);
  const double u = 1.1102230246251565e-16;                              // 2^-53
  const bool live = (s_live[cl] != 0u);
  const double* rc = r + (size_t)cl * O1;
  const double r0 = live ? rc[0] : 1.0;
  const uint32_t len = s_len[cl];
  const bool valid = live && len >= order && r0 > 2.0 * (double)FLT_EPSILON;   // (the reference zeroes the coefficients below FLT_EPSILON, src/SLAPredictor.c:274)
  const double delta = ((double)len * u) * r0 + (48.0 * u) * s_energy[cl];
  const double d = cert * (double)(2 * order + 1) * delta;
  double e_mine = nan, g_mine = 1.0;
  if (valid) {
    const double rb = (half == 0) ? (r0 + d) : (r0 - d);
    if (half == 0 || r0 - d > (double)FLT_EPSILON) { e_mine = schur_error<P>(rc, rb, order, g_mine); }
  }
  // the other end of my candidate's bracket sits 32 lanes away
  const double e_other = __hiloint2double(__shfl_xor(__double2hiint(e_mine), 32), __shfl_xor(__double2loint(e_mine), 32));
  const double g_other = __hiloint2double(__shfl_xor(__double2hiint(g_mine), 32), __shfl_xor(__double2loint(g_mine), 32));
  if (live) {
    double* o = out + ((uint64_t)slot0 + s) * O2;
    if (half == 0) {
      // slot layout of a certified candidate: { r0, width, log2(e_p / r0), 0, .. }
      const double e_hi = e_mine, e_lo = e_other;
      double w = inf, lg = 0.0;
      if (valid) {
        // (the share of the bound left for the rounding of the reference's Levinson-Durbin run: see k_search_finish)
        const double gmax = fmax(g_mine, g_other);
        const bool rounding_covered = ((double)(order + 2) * gmax * u * r0 <= (cert - 1.0) * (double)(2 * order + 1) * delta);
        if (rounding_covered && e_lo > 0.0 && e_hi >= e_lo && e_hi < inf) {
          const double lh = log2(e_hi / r0), ll = log2(e_lo / r0);
          w = 0.5 * (lh - ll) * 1.000001 + 1e-11;                       // (device log2: a few ulp)
          lg = 0.5 * (lh + ll);
        }
      }
      o[0] = rc[0];
      o[1] = (w == w) ? w : inf;
      o[2] = lg;
    }
    // the rest of the slot is zero: the two lanes of the candidate share the stores
    for (uint32_t k = 2 + half; k <= order; k += 2) { o[1 + k] = 0.0; }
  }
}

extern "C" uint32_t sla_hip_search_exact_lags(uint32_t order)
{
  const uint32_t nb = (order + 1 + 3) / 4;
  if (order < 1) { return 0; }
  return (nb <= 3) ? 12 : (nb <= 5) ? 20 : (nb <= 9) ? 36 : (nb <= 13) ? 52 : 0;
}

extern "C" int sla_hip_launch_search_exact_x(const int32_t* d_pcm, uint64_t plane_stride, uint32_t mid_side, uint32_t order,
                                           const sla_hip_lpc_group* d_groups, uint32_t num_groups, uint32_t max_window,
                                           uint32_t max_cands_per_group,
                                           const sla_hip_lpc_cand* d_cands, double* d_tile_sums, double* d_out,
                                           double exact_limit, double cert_safety, uint32_t* d_any_exact, sla_hip_stream_t stream, const sla_hip_launch_extra* extra)
{
  if (d_pcm == nullptr || d_groups == nullptr || d_cands == nullptr || d_tile_sums == nullptr || d_out == nullptr) { return SLA_APIRESULT_INVALID_ARGUMENT; }
  const uint32_t lags = sla_hip_search_exact_lags(order);
  if (lags == 0 || max_window > SLA_HIP_XTILE * SLA_HIP_XTILES) { return SLA_APIRESULT_EXCEED_HANDLE_CAPACITY; }
  if (max_window == 0 || max_cands_per_group == 0) { return SLA_APIRESULT_INVALID_ARGUMENT; }
  if (num_groups == 0) { return 0; }
  hipStream_t st = (hipStream_t)stream;
  clear_list cl = {{nullptr, nullptr, nullptr, nullptr}, {0, 0, 0, 0}};      // what the caller asked to have cleared on the way
  if (extra != nullptr) { for (int i = 0; i < 3; i++) { cl.ptr[i] = extra->clear_ptr[i]; cl.words[i] = extra->clear_words[i]; } }
  cl.ptr[3] = d_any_exact; cl.words[3] = (d_any_exact != nullptr) ? 1u : 0u;      // the flag k_search_finish raises
  const uint32_t tiles = (max_window + SLA_HIP_XTILE - 1) / SLA_HIP_XTILE;       // waves per group
  const uint32_t waves = num_groups * tiles;
  const dim3 grid((waves + 3) / 4), block(256);
  switch (lags) {
    case 12: hipLaunchKernelGGL(k_acf_tiles<3>, grid, block, 0, st, d_pcm, plane_stride, mid_side, d_groups, num_groups, tiles, d_tile_sums, cl); break;
    case 20: hipLaunchKernelGGL(k_acf_tiles_lds<5>, grid, block, 0, st, d_pcm, plane_stride, mid_side, d_groups, num_groups, tiles, d_tile_sums, cl); break;
    case 36: hipLaunchKernelGGL(k_acf_tiles_lds<9>, grid, block, 0, st, d_pcm, plane_stride, mid_side, d_groups, num_groups, tiles, d_tile_sums, cl); break;
    default: hipLaunchKernelGGL(k_acf_tiles_lds<13>, grid, block, 0, st, d_pcm, plane_stride, mid_side, d_groups, num_groups, tiles, d_tile_sums, cl); break;
  }
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) { return hip_rc(e); }
  // lanes = (group, candidate): `per` candidate slots per group and pass, `gpw` groups per wave
  const uint32_t per = (max_cands_per_group < XF_BATCH) ? max_cands_per_group : XF_BATCH;
  uint32_t gpw = XF_BATCH / per;
  if (gpw > XF_GROUPS) { gpw = XF_GROUPS; }
  if (gpw < 1) { gpw = 1; }
  const size_t lds_r = sizeof(double) * (size_t)gpw * per * (order + 1);
  // (the exact windows of mixed material run the recursion in registers as well: a launch that needs 70 KiB of LDS per
  // workgroup waits for CUs that the other streams' kernels have left that much of, even when -- loud material wider than
  // 16 bits -- it has nothing to do)
  // |x| < 2 in the search's unit (mid/side: side = l - r), so a window's energy stays below 4 x its length: when even
  // that is under the limit (16-bit material) every group takes the exact path and one launch does
  const bool all_exact = (exact_limit >= 4.0 * (double)max_window);
  const int pclass = (order <= 16) ? 16 : (order <= 32) ? 32 : (order <= 48) ? 48 : 64;      // (lags != 0: order <= 52)
#define SLA_FINISH(PP) do { \
    if (all_exact) { \
      e = ensure_dynamic_lds((const void*)k_search_finish<PP, 0>, lds_r); \
      if (e != hipSuccess) { return hip_rc(e); } \
      hipLaunchKernelGGL((k_search_finish<PP, 0>), dim3((num_groups + gpw - 1) / gpw), dim3(64), lds_r, st, order, lags, gpw, per, \
                         d_groups, num_groups, d_cands, d_tile_sums, d_out, exact_limit, cert_safety, d_any_exact); \
    } else { \
      e = ensure_dynamic_lds((const void*)k_search_finish<PP, 1>, lds_r); \
      if (e != hipSuccess) { return hip_rc(e); } \
      e = ensure_dynamic_lds((const void*)k_search_finish<PP, 2>, lds_r); \
      if (e != hipSuccess) { return hip_rc(e); } \
      if (cert_safety > 0.0) { \
        const uint64_t slots_bound = (uint64_t)num_groups * max_cands_per_group; \
        hipLaunchKernelGGL((k_search_cert<PP>), dim3((uint32_t)((slots_bound + XC_CANDS - 1) / XC_CANDS)), dim3(64), sizeof(double) * XC_CANDS * (order + 1), st, \
                           order, lags, max_cands_per_group, d_groups, num_groups, d_cands, d_tile_sums, d_out, exact_limit, cert_safety, d_any_exact); \
      } else { \
      hipLaunchKernelGGL((k_search_finish<PP, 1>), dim3((num_groups + gpw - 1) / gpw), dim3(64), lds_r, st, order, lags, gpw, per, \
                         d_groups, num_groups, d_cands, d_tile_sums, d_out, exact_limit, cert_safety, d_any_exact); \
      } \
      hipLaunchKernelGGL((k_search_finish<PP, 2>), dim3((num_groups + gpw - 1) / gpw), dim3(64), lds_r, st, order, lags, gpw, per, \
                         d_groups, num_groups, d_cands, d_tile_sums, d_out, exact_limit, cert_safety, d_any_exact); \
    } } while (0)
  switch (pclass) {
    case 16: SLA_FINISH(16); break;
    case 32: SLA_FINISH(32); break;
    case 48: SLA_FINISH(48); break;
    default: SLA_FINISH(64); break;
  }
#undef SLA_FINISH
  return hip_rc(hipGetLastError());
}

extern "C" int sla_hip_launch_search_exact(const int32_t* d_pcm, uint64_t plane_stride, uint32_t mid_side, uint32_t order,
                                           const sla_hip_lpc_group* d_groups, uint32_t num_groups, uint32_t max_window,
                                           uint32_t max_cands_per_group,
                                           const sla_hip_lpc_cand* d_cands, double* d_tile_sums, double* d_out,
                                           double exact_limit, double cert_safety, uint32_t* d_any_exact, sla_hip_stream_t stream)
{
  return sla_hip_launch_search_exact_x(d_pcm, plane_stride, mid_side, order, d_groups, num_groups, max_window, max_cands_per_group, d_cands, d_tile_sums, d_out, exact_limit, cert_safety, d_any_exact, stream, nullptr);
}

extern "C" int sla_hip_launch_plan(const sla_hip_lpc_group* d_groups, uint32_t num_superframes, uint32_t num_channels,
                                   uint32_t order, uint32_t bits_per_sample, const sla_hip_lpc_cand* d_cands,
                                   double* d_lpc_out, uint32_t* d_parts, uint32_t* d_num_parts, uint32_t* d_status,
                                   sla_hip_stream_t stream)
{
  if (d_groups == nullptr || d_cands == nullptr || d_lpc_out == nullptr || d_parts == nullptr || d_num_parts == nullptr
      || d_status == nullptr) { return SLA_APIRESULT_INVALID_ARGUMENT; }
  if (num_channels == 0 || num_channels > 8 || order < 1 || bits_per_sample == 0 || bits_per_sample > 32) { return SLA_APIRESULT_INVALID_ARGUMENT; }
  if (num_superframes == 0) { return 0; }
  hipLaunchKernelGGL(k_plan, dim3((num_superframes + 3) / 4), dim3(256), 0, (hipStream_t)stream, d_groups, num_superframes, num_channels,
                     order, bits_per_sample, d_cands, d_lpc_out, d_parts, d_num_parts, d_status,
                     (tuning().plan_margin > 0.0) ? tuning().plan_margin : PLAN_MARGIN);   /* tests raise it to force the host path */
  return hip_rc(hipGetLastError());
}

extern "C" int sla_hip_launch_expand(const sla_hip_superframe* d_superframes, uint32_t num_superframes,
                                     const uint32_t* d_parts, const uint32_t* d_num_parts, const uint32_t* d_status,
                                     uint32_t num_channels, uint32_t int_shift,
                                     const uint32_t* d_win_len, const uint32_t* d_win_off, uint32_t num_windows,
                                     uint32_t* d_run, uint32_t* d_prefix, sla_hip_lpc_group* d_groups, sla_hip_lpc_cand* d_cands,
                                     sla_hip_acf_job* d_acf_jobs, uint32_t group_capacity,
                                     uint32_t* counts, uint32_t sequence, sla_hip_stream_t stream)
{
  if (d_superframes == nullptr || d_parts == nullptr || d_num_parts == nullptr || d_status == nullptr || d_run == nullptr
      || d_prefix == nullptr || d_groups == nullptr || d_cands == nullptr || d_acf_jobs == nullptr || counts == nullptr
      || (num_windows != 0 && (d_win_len == nullptr || d_win_off == nullptr))) { return SLA_APIRESULT_INVALID_ARGUMENT; }
  if (num_channels == 0 || num_channels > 8 || int_shift > 31) { return SLA_APIRESULT_INVALID_ARGUMENT; }
  hipLaunchKernelGGL(k_expand_scan, dim3(1), dim3(EXPAND_THREADS), 0, (hipStream_t)stream, d_superframes, num_superframes, d_parts,
                     d_num_parts, d_status, num_channels, d_win_len, num_windows, d_run, d_prefix, group_capacity,
                     (volatile uint32_t*)counts, sequence);
  if (num_superframes != 0) {
    hipLaunchKernelGGL(k_expand_write, dim3((num_superframes + 255) / 256), dim3(256), 0, (hipStream_t)stream, d_superframes, num_superframes,
                       d_parts, d_num_parts, num_channels, int_shift, d_win_len, d_win_off, num_windows, d_run, d_prefix, d_groups, d_cands,
                       d_acf_jobs);
  }
  return hip_rc(hipGetLastError());
}

extern "C" int sla_hip_launch_lattice(const int32_t* d_pcm, uint64_t plane_stride, uint32_t mid_side, uint32_t order,
                                      const sla_hip_lattice_chunk* d_chunks, uint32_t num_chunks,
                                      const int32_t* d_kint, int32_t* d_residual, sla_hip_stream_t stream)
{
  if (d_pcm == nullptr || d_chunks == nullptr || d_kint == nullptr || d_residual == nullptr) { return SLA_APIRESULT_INVALID_ARGUMENT; }
  if (order < 1 || order > 255 || (order + LAT_T - 1) / LAT_T >= 32) { return SLA_APIRESULT_INVALID_ARGUMENT; }
  if (num_chunks == 0) { return 0; }
  hipLaunchKernelGGL(k_lattice, dim3((num_chunks + 3) / 4), dim3(256), 0, (hipStream_t)stream, d_pcm, plane_stride,
                     mid_side, order, d_chunks, num_chunks, d_kint, d_residual, (unsigned long long*)nullptr, tuning().lattice_plain ? 2u : 0u);
  return hip_rc(hipGetLastError());
}

extern "C" int sla_hip_launch_lattice_groups_x(const int32_t* d_pcm, uint64_t plane_stride, uint32_t mid_side, uint32_t order,
                                             const sla_hip_lpc_group* d_groups, uint32_t num_groups, uint32_t max_window,
                                             const int32_t* d_kint, int32_t* d_residual, sla_hip_stream_t stream, const sla_hip_launch_extra* extra)
{
  if (d_pcm == nullptr || d_groups == nullptr || d_kint == nullptr || d_residual == nullptr) { return SLA_APIRESULT_INVALID_ARGUMENT; }
  if (order < 1 || order > 255 || (order + LAT_T - 1) / LAT_T >= 32 || max_window == 0) { return SLA_APIRESULT_INVALID_ARGUMENT; }
  if (num_groups == 0) { return 0; }
  const uint32_t per = (SLA_WAVE - (order + LAT_T - 1) / LAT_T) * LAT_T;
  const uint32_t cpg = (max_window + per - 1) / per;
  const uint64_t waves = (uint64_t)num_groups * cpg;
  if (waves > 0x7FFFFFFFull) { return SLA_APIRESULT_EXCEED_HANDLE_CAPACITY; }
  hipLaunchKernelGGL(k_lattice_groups, dim3((uint32_t)((waves + 3) / 4)), dim3(256), 0, (hipStream_t)stream, d_pcm, plane_stride,
                     mid_side, order, d_groups, num_groups, cpg, d_kint, d_residual, span_of(extra), tuning().lattice_plain ? 2u : 0u);
  return hip_rc(hipGetLastError());
}

extern "C" int sla_hip_launch_lattice_groups(const int32_t* d_pcm, uint64_t plane_stride, uint32_t mid_side, uint32_t order,
                                             const sla_hip_lpc_group* d_groups, uint32_t num_groups, uint32_t max_window,
                                             const int32_t* d_kint, int32_t* d_residual, sla_hip_stream_t stream)
{
  return sla_hip_launch_lattice_groups_x(d_pcm, plane_stride, mid_side, order, d_groups, num_groups, max_window, d_kint, d_residual, stream, nullptr);
}

extern "C" uint32_t sla_hip_lattice_chunk_samples(uint32_t order)
{
  return (SLA_WAVE - (order + LAT_T - 1) / LAT_T) * LAT_T;
}

extern "C" int sla_hip_launch_lattice_raw(const int32_t* d_samples, uint64_t plane_stride, uint32_t order,
                                          const sla_hip_lattice_chunk* d_chunks, uint32_t num_chunks,
                                          const int32_t* d_kint, int32_t* d_residual, sla_hip_stream_t stream)
{
  if (d_samples == nullptr || d_chunks == nullptr || d_kint == nullptr || d_residual == nullptr) { return SLA_APIRESULT_INVALID_ARGUMENT; }
  if (order < 1 || order > 255 || (order + LAT_T - 1) / LAT_T >= 32) { return SLA_APIRESULT_INVALID_ARGUMENT; }
  if (num_chunks == 0) { return 0; }
  hipLaunchKernelGGL(k_lattice, dim3((num_chunks + 3) / 4), dim3(256), 0, (hipStream_t)stream, d_samples, plane_stride,
                     0u, order, d_chunks, num_chunks, d_kint, d_residual, (unsigned long long*)nullptr, 1u | (tuning().lattice_plain ? 2u : 0u));
  return hip_rc(hipGetLastError());
}

extern "C" int sla_hip_launch_lpc_f64(const double* d_samples, uint32_t order,
                                      const sla_hip_lpc_group* d_groups, uint32_t num_groups, uint32_t max_window,
                                      uint32_t max_cands_per_group, const sla_hip_lpc_cand* d_cands,
                                      double* d_out, sla_hip_stream_t stream)
{
  return launch_lpc_impl(reinterpret_cast<const int32_t*>(d_samples), 0, 0, order, d_groups, num_groups, max_window, max_cands_per_group,
                         d_cands, nullptr, d_out, nullptr, nullptr, nullptr, stream, 512u, nullptr, nullptr);
}

extern "C" int sla_hip_launch_emphasis_i32(const int32_t* d_in, int32_t* d_out, uint32_t num_samples, int32_t previous,
                                           uint32_t coef_shift, sla_hip_stream_t stream)
{
  if (d_in == nullptr || d_out == nullptr || coef_shift == 0 || coef_shift > 30) { return SLA_APIRESULT_INVALID_ARGUMENT; }
  if (num_samples == 0) { return 0; }
  hipLaunchKernelGGL(k_emphasis_i32, dim3((num_samples + 255) / 256), dim3(256), 0, (hipStream_t)stream, d_in, d_out, num_samples, previous, coef_shift);
  return hip_rc(hipGetLastError());
}

extern "C" int sla_hip_launch_emphasis_f64(const double* d_in, double* d_out, uint32_t num_samples, uint32_t coef_shift,
                                           sla_hip_stream_t stream)
{
  if (d_in == nullptr || d_out == nullptr || coef_shift == 0 || coef_shift > 30) { return SLA_APIRESULT_INVALID_ARGUMENT; }
  if (num_samples == 0) { return 0; }
  hipLaunchKernelGGL(k_emphasis_f64, dim3((num_samples + 255) / 256), dim3(256), 0, (hipStream_t)stream, d_in, d_out, num_samples, coef_shift);
  return hip_rc(hipGetLastError());
}

static int launch_tail_impl(const int32_t* d_res_in, int32_t* d_res_out, uint64_t plane_stride,
                            const sla_hip_tail_job* d_jobs, uint32_t num_jobs, uint32_t longterm_order,
                            uint32_t lms_order, uint64_t* d_fold_sum, sla_hip_stream_t stream, uint32_t stage_flags,
                            unsigned long long* span = nullptr);

extern "C" int sla_hip_launch_tail_x(const int32_t* d_res_in, int32_t* d_res_out, uint64_t plane_stride,
                                     const sla_hip_tail_job* d_jobs, uint32_t num_jobs, uint32_t longterm_order,
                                     uint32_t lms_order, uint64_t* d_fold_sum, sla_hip_stream_t stream, const sla_hip_launch_extra* extra)
{
  return launch_tail_impl(d_res_in, d_res_out, plane_stride, d_jobs, num_jobs, longterm_order, lms_order, d_fold_sum, stream, 0u, span_of(extra));
}

extern "C" int sla_hip_launch_tail(const int32_t* d_res_in, int32_t* d_res_out, uint64_t plane_stride,
                                   const sla_hip_tail_job* d_jobs, uint32_t num_jobs, uint32_t longterm_order,
                                   uint32_t lms_order, uint64_t* d_fold_sum, sla_hip_stream_t stream)
{
  return launch_tail_impl(d_res_in, d_res_out, plane_stride, d_jobs, num_jobs, longterm_order, lms_order, d_fold_sum, stream, 0u);
}

extern "C" int sla_hip_launch_tail_stages(const int32_t* d_res_in, int32_t* d_res_out, uint64_t plane_stride,
                                          const sla_hip_tail_job* d_jobs, uint32_t num_jobs, uint32_t longterm_order,
                                          uint32_t lms_order, uint32_t skip_lms, uint64_t* d_fold_sum, sla_hip_stream_t stream)
{
  return launch_tail_impl(d_res_in, d_res_out, plane_stride, d_jobs, num_jobs, longterm_order, lms_order, d_fold_sum, stream,
                          skip_lms ? 1u : 0u);
}

/* one-tap-per-lane waves of k_tailk beyond which two taps per lane are chosen (1024 SIMDs) */
#define TAILK2_WAVES 2048u
static int launch_tail_impl(const int32_t* d_res_in, int32_t* d_res_out, uint64_t plane_stride,
                            const sla_hip_tail_job* d_jobs, uint32_t num_jobs, uint32_t longterm_order,
                            uint32_t lms_order, uint64_t* d_fold_sum, sla_hip_stream_t stream, uint32_t stage_flags,
                            unsigned long long* span)
{
  if (d_res_in == nullptr || d_res_out == nullptr || d_jobs == nullptr || d_fold_sum == nullptr) { return SLA_APIRESULT_INVALID_ARGUMENT; }
  if (longterm_order > 5 || !(longterm_order & 1)) { return SLA_APIRESULT_INVALID_ARGUMENT; }
  if (!(lms_order == 4 || lms_order == 8 || lms_order == 16 || lms_order == 32)) { return SLA_APIRESULT_EXCEED_HANDLE_CAPACITY; }
  if (num_jobs == 0) { return 0; }
  const uint32_t tw = tuning().tail_waves;
  hipStream_t st = (hipStream_t)stream;
  /* K taps of each history per lane (k_tailk).  Measured (tests/tools/chunk_sweep.py, LMS order 8, ms per launch; k_tail /
   * k_tail2 = the 2 * order- and order-lanes-per-job kernels of rounds 1 - 2, deleted in round 4):
   *   jobs     blocks          k_tail  k_tail2   K = 1   K = 2   K = 4
   *    3.5 k   4096 (C2 300 s)   0.42    0.49     0.35    0.43    0.62
   *    7 k     4096 (C2)         0.44    0.57     0.37    0.43    0.62
   *   14 k     4096 (C3 600 s)   0.99    0.78     0.50    0.49    0.68
   *   11 k     8192 (C5 120 s)   1.26    1.46     0.96    0.95    1.33
   *   90 k     4096 (C3)         4.33    3.60     1.95    1.46    1.55
   *  169 k     8192 (C5)        16.5    13.2      7.45    5.45    6.70
   * One tap per lane has the shortest chain per sample (0.35 ms per 4096 samples) and holds up to about two waves per SIMD;
   * beyond that two taps per lane halve the work per job.  Four never won: its 16-sample runs per lane cost more in the
   * long-term stage and registers than the shared per-sample work saves.  (One LANE per job -- histories in registers,
   * tiles transposed through LDS, round 3 -- lost to all of them: 16 half-rate 32-bit products per sample in one lane's
   * chain.) */
  uint32_t k = tuning().tail_taps;                                     /* 0 = by the number of jobs */
  const uint32_t k1_waves = (num_jobs * lms_order + 63u) / 64u;        /* waves of K = 1 */
  if (k != 1 && k != 2 && k != 4) { k = (lms_order > 16 || (k1_waves > TAILK2_WAVES && lms_order >= 4)) ? 2u : 1u; }
  if (k == 1 && lms_order > 16) { k = 2; }                /* at most sixteen lanes per job */
  if (k > lms_order / 2) { k = lms_order / 2; }          /* at least two */
  {
    /* four waves per workgroup: one workgroup fills a CU's four SIMDs with one wave each (one-wave workgroups landed two
     * on a SIMD while other CUs stood empty: C2 0.44 against 0.37 ms, C3 1.83 against 1.46) */
    const uint32_t twk = (tw >= 1 && tw <= 4) ? tw : 4u;
    const uint32_t jpw = 64 / (lms_order / k);
    const uint32_t jpb = twk * jpw;
    dim3 gridk((num_jobs + jpb - 1) / jpb), blockk(64 * twk);
#define LAUNCH_TAILK(O, KK) hipLaunchKernelGGL((k_tailk<O, KK>), gridk, blockk, 0, st, d_res_in, d_res_out, plane_stride, d_jobs, num_jobs, longterm_order, d_fold_sum, span, stage_flags)
    switch (lms_order * 16 + k) {
      case 4 * 16 + 1:  LAUNCH_TAILK(4, 1); break;
      case 8 * 16 + 1:  LAUNCH_TAILK(8, 1); break;
      case 16 * 16 + 1: LAUNCH_TAILK(16, 1); break;
      case 4 * 16 + 2:  LAUNCH_TAILK(4, 2); break;
      case 8 * 16 + 2:  LAUNCH_TAILK(8, 2); break;
      case 8 * 16 + 4:  LAUNCH_TAILK(8, 4); break;
      case 16 * 16 + 2: LAUNCH_TAILK(16, 2); break;
      case 16 * 16 + 4: LAUNCH_TAILK(16, 4); break;
      case 32 * 16 + 2: LAUNCH_TAILK(32, 2); break;
      case 32 * 16 + 4: LAUNCH_TAILK(32, 4); break;
      default: return SLA_APIRESULT_EXCEED_HANDLE_CAPACITY;
    }
#undef LAUNCH_TAILK
  }
  return hip_rc(hipGetLastError());
}

extern "C" int sla_hip_launch_ltm_acf_x(const int32_t* d_residual, uint64_t plane_stride,
                                      const sla_hip_acf_job* d_jobs, uint32_t num_jobs, uint32_t fft_size,
                                      const double* d_twiddles, double* d_scratch, uint32_t scratch_slots,
                                      double* d_acf_head, uint32_t head, sla_hip_stream_t stream, const sla_hip_launch_extra* extra)
{
  if (d_residual == nullptr || d_jobs == nullptr || d_twiddles == nullptr || d_acf_head == nullptr) { return SLA_APIRESULT_INVALID_ARGUMENT; }
  if (fft_size < 1024 || fft_size > 65536 * 2 || (fft_size & (fft_size - 1)) || head == 0 || head > fft_size) { return SLA_APIRESULT_INVALID_ARGUMENT; }
  if (num_jobs == 0) { return 0; }
  uint32_t log2F = 0;
  while ((1u << log2F) < fft_size) { log2F++; }
  hipStream_t st = (hipStream_t)stream;
  const size_t lds = sizeof(double) * (size_t)fft_size;
  unsigned long long* span = span_of(extra);
  if (lds <= SLA_HIP_LDS_BUDGET && log2F >= 12 && log2F <= 14) {
    // the capacities the encoder is created with (2048 .. 8192 samples per block): fewer LDS passes, same bits
    hipError_t e = hipSuccess;
#define SLA_ACF2(LL, TT, REC) do { \
      e = ensure_dynamic_lds((const void*)k_ltm_acf2<LL, TT, REC>, lds); \
      if (e != hipSuccess) { return hip_rc(e); } \
      hipLaunchKernelGGL((k_ltm_acf2<LL, TT, REC>), dim3(num_jobs), dim3(TT), lds, st, d_residual, plane_stride, d_jobs, num_jobs, d_twiddles, d_acf_head, head, span); } while (0)
    const bool rec = (head == SLA_HIP_ACF_RECORD);
    // 64 KiB of LDS or less: 512 threads, two workgroups per CU; a 16384-point job owns the CU's LDS: 1024 threads (measured
    // on 11250 such jobs: 0.91 ms against 1.10 ms with 512 threads; k_ltm_acf: 1.85 ms) -- four waves per SIMD either way
    if (log2F == 12) { if (rec) { SLA_ACF2(11, 512, true); } else { SLA_ACF2(11, 512, false); } }
    else if (log2F == 13) { if (rec) { SLA_ACF2(12, 512, true); } else { SLA_ACF2(12, 512, false); } }
    else { if (rec) { SLA_ACF2(13, 1024, true); } else { SLA_ACF2(13, 1024, false); } }
#undef SLA_ACF2
  } else if (lds <= SLA_HIP_LDS_BUDGET) {
    hipError_t e = ensure_dynamic_lds((const void*)k_ltm_acf<true>, lds);
    if (e != hipSuccess) { return hip_rc(e); }
    hipLaunchKernelGGL(k_ltm_acf<true>, dim3(num_jobs), dim3(ACF_THREADS), lds, st, d_residual, plane_stride, d_jobs, num_jobs,
                       log2F, d_twiddles, (double*)nullptr, d_acf_head, head, span);
  } else {
    if (d_scratch == nullptr || scratch_slots == 0) { return SLA_APIRESULT_INVALID_ARGUMENT; }
    uint32_t grid = (num_jobs < scratch_slots) ? num_jobs : scratch_slots;
    hipLaunchKernelGGL(k_ltm_acf<false>, dim3(grid), dim3(ACF_THREADS), 0, st, d_residual, plane_stride, d_jobs, num_jobs,
                       log2F, d_twiddles, d_scratch, d_acf_head, head, span);
  }
  return hip_rc(hipGetLastError());
}

extern "C" int sla_hip_launch_ltm_acf(const int32_t* d_residual, uint64_t plane_stride,
                                      const sla_hip_acf_job* d_jobs, uint32_t num_jobs, uint32_t fft_size,
                                      const double* d_twiddles, double* d_scratch, uint32_t scratch_slots,
                                      double* d_acf_head, uint32_t head, sla_hip_stream_t stream)
{
  return sla_hip_launch_ltm_acf_x(d_residual, plane_stride, d_jobs, num_jobs, fft_size, d_twiddles, d_scratch, scratch_slots, d_acf_head, head, stream, nullptr);
}

extern "C" int sla_hip_launch_ltm_solve(const double* d_acf_records, const sla_hip_lpc_group* d_groups, uint32_t num_jobs,
                                        uint32_t longterm_order, sla_hip_tail_job* d_jobs, sla_hip_stream_t stream)
{
  if (d_acf_records == nullptr || d_groups == nullptr || d_jobs == nullptr) { return SLA_APIRESULT_INVALID_ARGUMENT; }
  if (longterm_order == 0 || longterm_order > LTM_NT || (longterm_order & 1u) == 0) { return SLA_APIRESULT_INVALID_ARGUMENT; }
  if (num_jobs == 0) { return 0; }
  const dim3 grid((num_jobs + 63) / 64), block(64);
  if (longterm_order == 1) { hipLaunchKernelGGL(k_ltm_solve<1>, grid, block, 0, (hipStream_t)stream, d_acf_records, d_groups, num_jobs, d_jobs); }
  else if (longterm_order == 3) { hipLaunchKernelGGL(k_ltm_solve<3>, grid, block, 0, (hipStream_t)stream, d_acf_records, d_groups, num_jobs, d_jobs); }
  else { hipLaunchKernelGGL(k_ltm_solve<5>, grid, block, 0, (hipStream_t)stream, d_acf_records, d_groups, num_jobs, d_jobs); }
  return hip_rc(hipGetLastError());
}

// ---------------------------------------------------------------------------------------------
// Bit-pack on the device: recursive-Rice / Golomb / gamma coding of the final
// residual (reference src/SLACoder.c:45-82, 120-138, 224-270, 429-467), block assembly and CRC16
// (src/SLAEncoder.c:682-798, src/SLAUtility.c:322-339).
//
//   k_rice_k     one lane per (block, channel): walks the residual with the two adaptive parameters
//                (8.8 fixed-point EMA, serial in time) and stores log2 of both Rice moduli per sample.
//                Blocks in fixed-Golomb mode need no state.
//   k_rice_bits  one wave per (block, channel): the channel's total bit count (code lengths are stateless)
//   k_rice_write one workgroup per block: header bytes, then tiles of 256 interleaved (sample, channel)
//                elements -- code length from (value, k0, k1), workgroup prefix sum -> bit offset, and
//                the <= 3 non-zero pieces of the codeword OR-ed into the zero-initialised image
//                (MSB-first, 32-bit atomics on byte-swapped words).  Unary zero runs cost nothing.
//   k_block_crc  one wave per block: slice-parallel CRC16-IBM over the block, size + CRC patched into the header.
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t fold_u32(int32_t s) { const uint32_t u = (uint32_t)s << 1; return (s < 0) ? ~u : u; }
__device__ __forceinline__ uint32_t ceil_log2_u32(uint32_t x) { return (x > 1) ? (32u - (uint32_t)__builtin_clz(x - 1u)) : 0u; }

// log2 of the Rice modulus of an adaptive parameter: 2^ceil(log2(round(p/2))), at least 1   src/SLACoder.c:30-31
__device__ __forceinline__ uint32_t rice_k(uint64_t p)
{
  uint32_t v = (uint32_t)(((p >> 1) + 128u) >> 8);
  v = v ? v : 1u;
  return ceil_log2_u32(v);
}
// 119/128 old + 9/128 code, the code term in 32-bit wrapping arithmetic                         src/SLACoder.c:26-28
__device__ __forceinline__ uint64_t rice_adapt(uint64_t p, uint32_t code)
{
  return (119u * p + (uint64_t)(uint32_t)(9u * (uint32_t)(code << 8)) + 64u) >> 7;
}
__device__ __forceinline__ uint32_t gamma_len(uint32_t g) { return (g == 0) ? 1u : (2u * ceil_log2_u32(g + 2u) - 1u); }

// length in bits of the recursive-Rice codeword of `v` under moduli 2^k0, 2^k1 (two parameters)
__device__ __forceinline__ uint32_t rrice_len(uint32_t v, uint32_t k0, uint32_t k1)
{
  if (v < (1u << k0)) { return 1u + k0; }
  v -= (1u << k0);
  const uint32_t q = 1u + (v >> k1);
  return (q < 16u) ? (q + 1u + k1) : (17u + gamma_len(q - 16u) + k1);
}
// length of the Golomb codeword of v with modulus m                                             src/SLACoder.c:45-82
__device__ __forceinline__ uint32_t golomb_len(uint32_t v, uint32_t m)
{
  const uint32_t q = v / m, r = v - q * m;
  if ((m & (m - 1u)) == 0) { return q + 1u + ceil_log2_u32(m); }
  const uint32_t b = ceil_log2_u32(m), cut = (1u << b) - m;
  return q + 1u + ((r < cut) ? (b - 1u) : b);
}

// The serial part only: the two adaptive parameters per sample (one lane per (block, channel)).  A lane issues its
// instructions one after the other, so every instruction that is not on the recurrence costs as much as one that is:
// the code lengths -- a function of (value, k0, k1), no state -- are left to k_rice_bits (a 10-second clip: 0.83 ->
// see DESIGN.md ms for the two kernels).
// One step of the walk in 32-bit arithmetic.  A parameter never leaves 32 bits (it starts as (uint32)(init << 8) and
// every step maps p to < 119/128 p + 2^25), but 119 p does: with p = 128 q + r and c = 128 cq + cr
//   (119 p + c + 64) >> 7  =  119 q + cq + ((119 r + cr + 64) >> 7),
// every term below 2^32.  c = (uint32)(9 (code << 8)) depends on the sample alone, so only the four operations of the
// second line sit on the serial chain.
__device__ __forceinline__ uint32_t rice_adapt32(uint32_t p, uint32_t code)
{
  // (measured: spelling 9 x and 119 q as opaque shift-adds instead of the multiplies the compiler picks is slower)
  const uint32_t c = 9u * (code << 8);
  const uint32_t q = p >> 7, r = p & 127u;
  return 119u * q + (c >> 7) + ((119u * r + (c & 127u) + 64u) >> 7);
}
__device__ __forceinline__ uint32_t rice_k32(uint32_t p)
{
  uint32_t v = ((p >> 1) + 128u) >> 8;
  v = v ? v : 1u;
  return ceil_log2_u32(v);
}

__global__ __launch_bounds__(64)
void k_rice_k(const int32_t* __restrict__ res, uint64_t stride, const sla_hip_rice_job* __restrict__ jobs,
              uint32_t num_jobs, uint16_t* __restrict__ kk)
{
  const uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= num_jobs) { return; }
  const sla_hip_rice_job job = jobs[j];
  if (job.golomb_m != 0) { return; }                         // fixed-parameter mode: stateless
  const int32_t* in = res + (uint64_t)job.channel * stride + job.blk_off;
  uint16_t* ko = kk + (uint64_t)job.channel * stride + job.blk_off;
  const uint32_t n = job.blk_len;
  uint32_t p0 = (uint32_t)(job.rice_init << 8), p1 = p0;
  // one sample: both moduli as they stand, then the adaptation (the second parameter only when the first stage passes
  // the value on -- a select, the lanes of a wave walk different blocks)                          src/SLACoder.c:120-138
  auto step = [&](int32_t sample) -> uint32_t {
    const uint32_t v = fold_u32(sample);
    const uint32_t k0 = rice_k32(p0), k1 = rice_k32(p1);
    const uint32_t m0 = 1u << k0;
    const uint32_t n1 = rice_adapt32(p1, v - m0);
    p0 = rice_adapt32(p0, v);
    p1 = (v >= m0) ? n1 : p1;
    return k0 | (k1 << 8);
  };
  uint32_t s = 0;
  // up to the first sample whose residual and k addresses are both 16-byte aligned (block starts usually are)
  while (s < n && (((uintptr_t)(in + s) & 15u) != 0 || ((uintptr_t)(ko + s) & 15u) != 0) && s < 8) { ko[s] = (uint16_t)step(in[s]); s++; }
  if ((((uintptr_t)(in + s) & 15u) == 0) && (((uintptr_t)(ko + s) & 15u) == 0)) {
    for (; s + 8 <= n; s += 8) {
      const int4 a = *reinterpret_cast<const int4*>(in + s);          // a lane reads its own line: 16 bytes per access,
      const int4 b = *reinterpret_cast<const int4*>(in + s + 4);      // not 4, and one 16-byte store for 8 samples
      uint4 o;
      o.x = step(a.x); o.x |= step(a.y) << 16;
      o.y = step(a.z); o.y |= step(a.w) << 16;
      o.z = step(b.x); o.z |= step(b.y) << 16;
      o.w = step(b.z); o.w |= step(b.w) << 16;
      *reinterpret_cast<uint4*>(ko + s) = o;
    }
  }
  for (; s < n; s++) { ko[s] = (uint16_t)step(in[s]); }
}

// ---------------------------------------------------------------------------------------------
// k_rice_k2: the same walk as a two-lane pipeline, for files with few (block, channel) jobs -- where k_rice_k's duration
// is ONE job's serial walk (~105 ns per sample: a ten-second clip waits 0.43 ms for it with the chip empty).
// What is serial in the walk is only the two recurrences p0' = adapt(p0, v) and p1' = v >= 2^k0 ? adapt(p1, v - 2^k0) : p1;
// everything around them is a function of one sample: the code terms c = 9 (code << 8) split into c >> 7 and (c & 127) + 64,
// the exponents k = rice_k(p), the comparison.  So, per batch of 64 samples of a job:
//   P1  all lanes, lane = sample: fold the residual, the first recurrence's two code terms -> LDS
//   S   lane A walks p0 over batch t while lane B walks p1 over batch t - 1 -- the SAME instruction stream (B's update is
//       conditional, A's condition is always true): q = p >> 7, r = p & 127, p' = 119 q + hi + ((119 r + lo) >> 7);
//       both leave the parameter AS IT STOOD before each sample in LDS
//   P2  all lanes: k0 = rice_k(p0), the condition v >= 2^k0 and the second recurrence's code terms of v - 2^k0 -> LDS
//   P3  all lanes, batch t - 1: k1 = rice_k(p1), k0 | k1 << 8 -> global memory
// Eight jobs per wave (sixteen lanes in S).  Same values as k_rice_k, sample for sample (test_rice_walk_kernels_agree and
// every byte comparison of the suite: the pack stage chooses by the number of jobs).
// ---------------------------------------------------------------------------------------------
#define RK2_JOBS 8
// (Rows of 65 entries: in S sixteen lanes walk sixteen different rows in lock-step.  With rows of 64 every lane's operand sat on
// the same banks -- a 16-way conflict on each read, 8-way on each store: 6.0 conflict cycles per LDS instruction,
// profiles/r3_sq_counters_c2.csv.  Now row j of in_a starts at bank 2 j, of in_b at 16 + 2 j; of p0 at j, of p1 at 8 + j.)
struct rk2_lds {
  uint32_t in_a[RK2_JOBS][65][2];      // first recurrence: c >> 7, ((c & 127) + 64) | 1 << 8
  uint32_t in_b[RK2_JOBS][65][2];      // second recurrence: the same of v - 2^k0, bit 8 = the update happens
  uint32_t p0[RK2_JOBS][65], p1[RK2_JOBS][65];      // the parameters before each sample
  uint32_t v[RK2_JOBS][64];            // folded residual
  uint32_t k0[2][RK2_JOBS][64];        // first exponent, kept for one more step (the second one is a batch behind)
};

__global__ __launch_bounds__(256)
void k_rice_k2(const int32_t* __restrict__ res, uint64_t stride, const sla_hip_rice_job* __restrict__ jobs,
               uint32_t num_jobs, uint16_t* __restrict__ kk)
{
  __shared__ rk2_lds s_all[4];
  const uint32_t lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  rk2_lds& L = s_all[wv];
  const uint32_t wave = blockIdx.x * 4 + wv;
  // lane l holds the fields of the wave's job l & 7; a phase that works on job j fetches them with a shuffle
  const uint32_t jb = wave * RK2_JOBS + (lane & 7u);
  const bool have = (jb < num_jobs);
  const sla_hip_rice_job job = jobs[have ? jb : 0];
  const uint32_t my_n = (have && job.golomb_m == 0) ? job.blk_len : 0u;          // fixed-parameter mode: stateless, nothing to walk
  const uint64_t my_off = (uint64_t)job.channel * stride + job.blk_off;
  const uint32_t nmax = umax_wave(my_n);
  const uint32_t nb = (nmax + 63u) / 64u;
  // the sixteen walking lanes: lane 2 j is A of job j, lane 2 j + 1 its B
  const uint32_t js = (lane >> 1) & 7u, role = lane & 1u;
  uint32_t p = (uint32_t)__shfl((int)(uint32_t)(job.rice_init << 8), (int)js);
  // the eight jobs' lengths and plane offsets as wave-uniform values (scalar registers), fetched once
  uint32_t n_of[RK2_JOBS]; uint64_t off_of[RK2_JOBS];
#pragma unroll
  for (int j = 0; j < RK2_JOBS; j++) {
    n_of[j] = (uint32_t)__builtin_amdgcn_readlane((int)my_n, j);
    off_of[j] = ((uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(my_off >> 32), j) << 32)
              | (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)my_off, j);
  }
  // residuals travel one batch ahead of their use: batch t + 1 is requested before the walk over batch t starts
  int32_t ahead[RK2_JOBS];
#pragma unroll
  for (int j = 0; j < RK2_JOBS; j++) { ahead[j] = (lane < n_of[j]) ? res[off_of[j] + lane] : 0; }
  for (uint32_t t = 0; t <= nb; t++) {
    if (t < nb) {
      // ---- P1: batch t of every job ----
#pragma unroll
      for (int j = 0; j < RK2_JOBS; j++) {
        const uint32_t v = fold_u32(ahead[j]);
        const uint32_t c = 9u * (v << 8);
        L.in_a[j][lane][0] = c >> 7; L.in_a[j][lane][1] = ((c & 127u) + 64u) | 256u;
        L.v[j][lane] = v;
      }
#pragma unroll
      for (int j = 0; j < RK2_JOBS; j++) {
        const uint32_t sidx = (t + 1u) * 64u + lane;
        ahead[j] = (sidx < n_of[j]) ? res[off_of[j] + sidx] : 0;
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    // ---- S: A on batch t, B on batch t - 1 ----
    if (lane < 2 * RK2_JOBS && (role ? (t >= 1u) : (t < nb))) {
      const uint32_t (*in)[2] = role ? L.in_b[js] : L.in_a[js];
      uint32_t* pout = role ? L.p1[js] : L.p0[js];
#pragma unroll 1
      for (uint32_t i0 = 0; i0 < 64; i0 += 8) {
        uint32_t hi[8], lo[8];
#pragma unroll
        for (int u = 0; u < 8; u++) { hi[u] = in[i0 + u][0]; lo[u] = in[i0 + u][1]; }      // eight steps' operands requested at once
#pragma unroll
        for (int u = 0; u < 8; u++) {
          pout[i0 + u] = p;
          const uint32_t q = p >> 7, r = p & 127u;
          const uint32_t np = 119u * q + hi[u] + ((119u * r + (lo[u] & 255u)) >> 7);
          p = (lo[u] & 256u) ? np : p;
        }
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    if (t < nb) {
      // ---- P2: batch t ----
#pragma unroll
      for (uint32_t j = 0; j < RK2_JOBS; j++) {
        const uint32_t k0 = rice_k32(L.p0[j][lane]);
        const uint32_t v = L.v[j][lane], m0 = 1u << k0;
        const uint32_t c = 9u * ((v - m0) << 8);
        L.in_b[j][lane][0] = c >> 7; L.in_b[j][lane][1] = ((c & 127u) + 64u) | ((v >= m0) ? 256u : 0u);
        L.k0[t & 1u][j][lane] = k0;
      }
    }
    if (t >= 1u) {
      // ---- P3: batch t - 1 ----
#pragma unroll
      for (int j = 0; j < RK2_JOBS; j++) {
        const uint32_t sidx = (t - 1u) * 64u + lane;
        const uint32_t k1 = rice_k32(L.p1[j][lane]);
        if (sidx < n_of[j]) { kk[off_of[j] + sidx] = (uint16_t)(L.k0[(t - 1u) & 1u][j][lane] | (k1 << 8)); }
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  }
}

// total body bits of every (block, channel): one wave per job, lanes stride over the samples
__global__ __launch_bounds__(256)
void k_rice_bits(const int32_t* __restrict__ res, uint64_t stride, const sla_hip_rice_job* __restrict__ jobs,
                 uint32_t num_jobs, const uint16_t* __restrict__ kk, uint64_t* __restrict__ chan_bits)
{
  const uint32_t lane = threadIdx.x & 63;
  const uint32_t j = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (j >= num_jobs) { return; }
  const sla_hip_rice_job job = jobs[j];
  const int32_t* in = res + (uint64_t)job.channel * stride + job.blk_off;
  const uint16_t* ki = kk + (uint64_t)job.channel * stride + job.blk_off;
  uint64_t bits = 0;
  if (job.golomb_m != 0) {
    for (uint32_t s = lane; s < job.blk_len; s += 64) { bits += golomb_len(fold_u32(in[s]), job.golomb_m); }
  } else {
    for (uint32_t s = lane; s < job.blk_len; s += 64) {
      const uint32_t k = ki[s];
      bits += rrice_len(fold_u32(in[s]), k & 0xFF, k >> 8);
    }
  }
  for (int off = 32; off > 0; off >>= 1) {
    const uint32_t lo = (uint32_t)__shfl_xor((int)(uint32_t)bits, off), hi = (uint32_t)__shfl_xor((int)(uint32_t)(bits >> 32), off);
    bits += ((uint64_t)hi << 32) | lo;
  }
  if (lane == 0) { chan_bits[j] = bits; }
}

// OR `len` (1..32) bits of `val` into the image, the first bit landing at absolute bit `pos` (MSB-first)
__device__ __forceinline__ void put_piece(uint32_t* __restrict__ img, uint64_t pos, uint32_t val, uint32_t len)
{
  const uint64_t w = pos >> 5;
  const uint32_t sh = (uint32_t)(pos & 31);
  const uint64_t v64 = ((uint64_t)val << (64 - len)) >> sh;          // bit 63 = first bit of word w
  const uint32_t hi = (uint32_t)(v64 >> 32), lo = (uint32_t)v64;
  if (hi) { atomicOr(&img[w], __builtin_bswap32(hi)); }
  if (lo) { atomicOr(&img[w + 1], __builtin_bswap32(lo)); }
}

__global__ __launch_bounds__(256)
void k_rice_write(const int32_t* __restrict__ res, const int32_t* __restrict__ pcm, uint64_t stride,
                  const uint16_t* __restrict__ kk, const sla_hip_pack_block* __restrict__ blocks,
                  const uint8_t* __restrict__ headers, uint32_t num_channels, uint32_t raw_shift, uint32_t mid_side,
                  uint32_t* __restrict__ img)
{
  __shared__ uint32_t s_wave[4];
  __shared__ uint64_t s_base;
  const sla_hip_pack_block b = blocks[blockIdx.x];
  const uint32_t C = num_channels, lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  // header bytes (sync, size/crc placeholders, sample count, type, per-channel fields), packed by the host
  for (uint32_t t = threadIdx.x; t < b.header_bytes; t += blockDim.x) {
    const uint64_t at = b.out_off + t;
    atomicOr(&img[at >> 2], (uint32_t)headers[b.header_off + t] << (8 * (at & 3)));
  }
  if (b.type == 1) { return; }                                         // SILENT: header only
  if (threadIdx.x == 0) { s_base = (b.out_off + b.header_bytes) * 8ull; }
  __syncthreads();
  const uint32_t total = b.num_samples * C;
  for (uint32_t e0 = 0; e0 < total; e0 += 256) {
    const uint32_t e = e0 + threadIdx.x;
    uint32_t len = 0, v = 0, k0 = 0, k1 = 0, m = 0;
    if (e < total) {
      const uint32_t s = e / C, c = e - s * C;
      if (b.type == 2) {                                               // RAW: fixed width per channel
        int32_t x;
        if (!mid_side) { x = pcm[(uint64_t)c * stride + b.blk_off + s] >> raw_shift; }
        else {
          const int32_t l = pcm[b.blk_off + s] >> raw_shift, r = pcm[stride + b.blk_off + s] >> raw_shift;
          x = (c == 0) ? ((int32_t)((uint32_t)l + (uint32_t)r) >> 1) : (int32_t)((uint32_t)l - (uint32_t)r);
        }
        v = fold_u32(x);
        len = b.raw_bits + ((c == 1 && mid_side) ? 1u : 0u);
      } else {
        v = fold_u32(res[(uint64_t)c * stride + b.blk_off + s]);
        m = b.golomb_m[c];
        if (m != 0) { len = golomb_len(v, m); }
        else {
          const uint32_t k = kk[(uint64_t)c * stride + b.blk_off + s];
          k0 = k & 0xFF; k1 = k >> 8;
          len = rrice_len(v, k0, k1);
        }
      }
    }
    // exclusive prefix sum of `len` over the 256 elements of the tile
    uint32_t inc = len;
    for (int off = 1; off < 64; off <<= 1) { const uint32_t o = __shfl_up(inc, off); if (lane >= (uint32_t)off) { inc += o; } }
    if (lane == 63) { s_wave[wv] = inc; }
    __syncthreads();
    uint32_t wave_base = 0, tile_total = 0;
    for (uint32_t w = 0; w < 4; w++) { const uint32_t t = s_wave[w]; wave_base += (w < wv) ? t : 0u; tile_total += t; }
    const uint64_t pos = s_base + wave_base + (inc - len);
    if (e < total) {
      if (b.type == 2) {
        if (len > 0) { put_piece(img, pos, (len >= 32) ? v : (v & ((1u << len) - 1u)), len); }
      } else if (m != 0) {                                             // Golomb: zeros(q) 1 payload
        const uint32_t q = v / m, r = v - q * m;
        uint32_t nb, pay;
        if ((m & (m - 1u)) == 0) { nb = ceil_log2_u32(m); pay = r; }
        else { const uint32_t bb = ceil_log2_u32(m), cut = (1u << bb) - m; if (r < cut) { nb = bb - 1; pay = r; } else { nb = bb; pay = r + cut; } }
        put_piece(img, pos + q, (1u << nb) | pay, nb + 1);
      } else if (v < (1u << k0)) {                                     // stage 0: 1 rest
        put_piece(img, pos, (1u << k0) | v, k0 + 1);
      } else {
        const uint32_t vv = v - (1u << k0), q = 1u + (vv >> k1), rest = vv & ((1u << k1) - 1u);
        if (q < 16u) {
          put_piece(img, pos + q, (1u << k1) | rest, k1 + 1);
        } else {                                                       // zeros(16) 1 gamma(q-16) rest
          const uint32_t g = q - 16u;
          put_piece(img, pos + 16, 1u, 1);
          uint64_t at = pos + 17;
          if (g == 0) { put_piece(img, at, 1u, 1); at += 1; }
          else { const uint32_t nd = ceil_log2_u32(g + 2u); put_piece(img, at + nd - 1, g + 1u, nd); at += 2 * nd - 1; }
          if (k1 > 0) { put_piece(img, at, rest, k1); }
        }
      }
    }
    __syncthreads();
    if (threadIdx.x == 0) { s_base += tile_total; }
    __syncthreads();
  }
}

// one wave per block (4 blocks per workgroup): slice-parallel CRC16, see sla_crc_dev.h
__global__ __launch_bounds__(256)
void k_block_crc(const sla_hip_pack_block* __restrict__ blocks, uint32_t num_blocks, uint32_t* __restrict__ img)
{
  __shared__ uint16_t table[256];
  crc16_build_table(table);
  __syncthreads();
  const uint32_t lane = threadIdx.x & 63;
  const uint32_t j = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (j >= num_blocks) { return; }
  const sla_hip_pack_block b = blocks[j];
  const uint32_t crc = crc16_wave((const uint8_t*)img, b.out_off + 8, b.out_off + b.out_bytes, table, lane);
  if (lane != 0) { return; }
  // size field = bytes after sync + size (32 bit, big endian), then the CRC (16 bit)
  const uint32_t sz = b.out_bytes - 6;
  const uint8_t patch[6] = { (uint8_t)(sz >> 24), (uint8_t)(sz >> 16), (uint8_t)(sz >> 8), (uint8_t)sz,
                             (uint8_t)(crc >> 8), (uint8_t)crc };
  for (uint32_t t = 0; t < 6; t++) {
    const uint64_t at = b.out_off + 2 + t;
    atomicOr(&img[at >> 2], (uint32_t)patch[t] << (8 * (at & 3)));
  }
}

#define RICE_K2_MAX_JOBS 32768u
extern "C" int sla_hip_launch_rice_len(const int32_t* d_residual, uint64_t plane_stride, const sla_hip_rice_job* d_jobs,
                                       uint32_t num_jobs, uint16_t* d_kk, uint64_t* d_chan_bits, sla_hip_stream_t stream)
{
  if (d_residual == nullptr || d_jobs == nullptr || d_kk == nullptr || d_chan_bits == nullptr) { return SLA_APIRESULT_INVALID_ARGUMENT; }
  if (num_jobs == 0) { return 0; }
  /* few jobs: the two-lane pipeline (k_rice_k2), whose walk per sample is a third of k_rice_k's; many: one lane per job,
   * a quarter of the instructions per job and sample.  tuning().rice_lanes: 1 / 2 force one or the other */
  const uint32_t rl = tuning().rice_lanes;
  if (rl == 2 || (rl == 0 && num_jobs <= RICE_K2_MAX_JOBS)) {
    hipLaunchKernelGGL(k_rice_k2, dim3((num_jobs + 4 * RK2_JOBS - 1) / (4 * RK2_JOBS)), dim3(256), 0, (hipStream_t)stream, d_residual, plane_stride,
                       d_jobs, num_jobs, d_kk);
  } else {
    hipLaunchKernelGGL(k_rice_k, dim3((num_jobs + 63) / 64), dim3(64), 0, (hipStream_t)stream, d_residual, plane_stride,
                       d_jobs, num_jobs, d_kk);
  }
  hipLaunchKernelGGL(k_rice_bits, dim3((num_jobs + 3) / 4), dim3(256), 0, (hipStream_t)stream, d_residual, plane_stride,
                     d_jobs, num_jobs, d_kk, d_chan_bits);
  return hip_rc(hipGetLastError());
}

extern "C" int sla_hip_launch_rice_write(const int32_t* d_residual, const int32_t* d_pcm, uint64_t plane_stride,
                                         const uint16_t* d_kk, const sla_hip_pack_block* d_blocks, uint32_t num_blocks,
                                         const uint8_t* d_headers, uint32_t num_channels, uint32_t raw_shift,
                                         uint32_t mid_side, uint32_t* d_image, sla_hip_stream_t stream)
{
  if (d_residual == nullptr || d_pcm == nullptr || d_kk == nullptr || d_blocks == nullptr || d_headers == nullptr || d_image == nullptr) {
    return SLA_APIRESULT_INVALID_ARGUMENT;
  }
  if (num_channels == 0 || num_channels > 8 || raw_shift > 31) { return SLA_APIRESULT_INVALID_ARGUMENT; }
  if (num_blocks == 0) { return 0; }
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(k_rice_write, dim3(num_blocks), dim3(256), 0, st, d_residual, d_pcm, plane_stride, d_kk, d_blocks,
                     d_headers, num_channels, raw_shift, mid_side, d_image);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) { return hip_rc(e); }
  hipLaunchKernelGGL(k_block_crc, dim3((num_blocks + 3) / 4), dim3(256), 0, st, d_blocks, num_blocks, d_image);
  return hip_rc(hipGetLastError());
}

// ---------------------------------------------------------------------------------------------
// k_unpack16: PCIe-side helper of SLAEncoder_EncodeWhole.  Input of <= 16 significant bits crosses the
// bus as int16 (the API's left-justified int32 has 16 zero low bits, checked on the host) and is
// re-expanded to the planar int32 layout every kernel reads.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256)
void k_unpack16(const int16_t* __restrict__ in, int32_t* __restrict__ out, uint64_t count)
{
  const uint64_t i = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) * 4;
  if (i + 4 <= count) {
    const short4 v = *reinterpret_cast<const short4*>(in + i);
    int32_t* o = out + i;
    o[0] = (int32_t)v.x << 16; o[1] = (int32_t)v.y << 16; o[2] = (int32_t)v.z << 16; o[3] = (int32_t)v.w << 16;
  } else {
    for (uint64_t k = i; k < count; k++) { out[k] = (int32_t)in[k] << 16; }
  }
}

// k_unpack24: the same for input of <= 24 significant bits that crossed the bus as three bytes per sample (option
// "upload24"): four samples = three 32-bit words.
__global__ __launch_bounds__(256)
void k_unpack24(const uint32_t* __restrict__ in, int32_t* __restrict__ out, uint64_t count)
{
  const uint64_t q = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;      // group of four samples
  const uint64_t i = q * 4;
  if (i + 4 <= count) {
    const uint32_t w0 = in[3 * q], w1 = in[3 * q + 1], w2 = in[3 * q + 2];
    int32_t* o = out + i;
    o[0] = (int32_t)(w0 << 8);
    o[1] = (int32_t)(((w0 >> 24) | (w1 << 8)) << 8);
    o[2] = (int32_t)(((w1 >> 16) | (w2 << 16)) << 8);
    o[3] = (int32_t)((w2 >> 8) << 8);
  } else if (i < count) {
    const uint8_t* b = reinterpret_cast<const uint8_t*>(in);
    for (uint64_t k = i; k < count; k++) { out[k] = (int32_t)(((uint32_t)b[3 * k] << 8) | ((uint32_t)b[3 * k + 1] << 16) | ((uint32_t)b[3 * k + 2] << 24)); }
  }
}

extern "C" int sla_hip_launch_unpack24(const uint8_t* d_in, int32_t* d_out, uint64_t count, sla_hip_stream_t stream)
{
  if (d_in == nullptr || d_out == nullptr) { return SLA_APIRESULT_INVALID_ARGUMENT; }
  if (count == 0) { return 0; }
  const uint64_t threads = (count + 3) / 4;
  hipLaunchKernelGGL(k_unpack24, dim3((uint32_t)((threads + 255) / 256)), dim3(256), 0, (hipStream_t)stream, (const uint32_t*)d_in, d_out, count);
  return hip_rc(hipGetLastError());
}

extern "C" int sla_hip_launch_unpack16(const int16_t* d_in, int32_t* d_out, uint64_t count, sla_hip_stream_t stream)
{
  if (d_in == nullptr || d_out == nullptr) { return SLA_APIRESULT_INVALID_ARGUMENT; }
  if (count == 0) { return 0; }
  const uint64_t threads = (count + 3) / 4;
  hipLaunchKernelGGL(k_unpack16, dim3((uint32_t)((threads + 255) / 256)), dim3(256), 0, (hipStream_t)stream, d_in, d_out, count);
  return hip_rc(hipGetLastError());
}
