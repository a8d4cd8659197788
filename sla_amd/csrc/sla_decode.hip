// sla_decode.hip -- gfx950 (MI355X / CDNA4) kernels of the SLA decode path.
//
// The decoder is the encoder's mirror: every stage is a recurrence in time (the entropy decoder's adaptive
// parameters, the sign-log LMS, the long-term synthesis, the IIR lattice, the de-emphasis), so parallelism
// exists only ACROSS blocks and channels -- and, inside one (block, channel), across the taps / lattice
// stages, which is what the lanes of a wave are used for.  All arithmetic is int32 / uint64 and wraps
// exactly like the reference's C.
//
// Kernel                 replaces (reference file:line)
//   k_dec_crc            src/SLAUtility.c:321-339 (CRC16 of each block), check at src/SLADecoder.c:343-352
//   k_dec_bits           src/SLADecoder.c:355-412 (block header fields), :440-481 (silent / raw / compressed body),
//                        src/SLACoder.c:84-118 (Golomb), :140-163 (gamma), :272-318 (recursive Rice), :469-506
//                        (SLACoder_GetDataArray), :406-427 (initial parameters)
//   k_dec_lms            src/SLAPredictor.c:1334-1463 (SLALMSFilter_SynthesizeInt32)
//   k_dec_ltm            src/SLAPredictor.c:1034-1119 (SLALongTermSynthesizer_SynthesizeInt32)
//   k_dec_lattice        src/SLAPredictor.c:610-740 (SLALPCSynthesizer_SynthesizeByParcorCoefInt32),
//                        :1768-1791 (SLAEmphasisFilter_DeEmphasisInt32)
//   k_dec_finish         src/SLAUtility.c:415-433 (mid/side -> left/right), src/SLADecoder.c:540-547 (left-justify)
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "sla_hip.h"
#include "sla_crc_dev.h"

namespace {

inline int hip_rc(hipError_t e) { return (e == hipSuccess) ? 0 : -(int)e; }

__device__ __forceinline__ int32_t unfold_i32(uint32_t u) { return (int32_t)(u >> 1) ^ -(int32_t)(u & 1u); }   // src/SLAUtility.h:39
__device__ __forceinline__ uint32_t ceil_log2_u32(uint32_t x) { return (x > 1) ? (32u - (uint32_t)__builtin_clz(x - 1u)) : 0u; }
// log2 of the Rice modulus of an adaptive parameter: 2^ceil(log2(round(p/2))), at least 1   src/SLACoder.c:30-31
__device__ __forceinline__ uint32_t rice_k(uint64_t p)
{
  uint32_t v = (uint32_t)(((p >> 1) + 128u) >> 8);
  v = v ? v : 1u;
  return ceil_log2_u32(v);
}
// 119/128 old + 9/128 code, the code term in 32-bit wrapping arithmetic                      src/SLACoder.c:26-28
__device__ __forceinline__ uint64_t rice_adapt(uint64_t p, uint32_t code)
{
  return (119u * p + (uint64_t)(uint32_t)(9u * (uint32_t)(code << 8)) + 64u) >> 7;
}
__device__ __forceinline__ uint32_t umax_wave(uint32_t v)
{
  for (int off = 32; off > 0; off >>= 1) { uint32_t o = __shfl_xor(v, off); v = (o > v) ? o : v; }
  return v;
}

// ---------------------------------------------------------------------------------------------
// MSB-first bit reader over the stream image (the file's bytes, viewed as 32-bit words).  Stateless apart
// from a bit position: a read is "the 32 bits at the position", assembled from two neighbouring words of
// the lane's row of an LDS window (one ds_read2 + a funnel shift), which the wave refills cooperatively
// (coalesced 256-byte loads, byte-swapped once) at every tile boundary.  No 64-bit shift register, no refill
// branches on the serial decode chain.  Words beyond the window (a tile that costs more than 64 bits per
// sample on average) come straight from memory.  Bytes past the end of the stream read as zero; a run of
// zeros that goes on past the end (only corrupt input does that) kills the reader: every later read returns 0.
// ---------------------------------------------------------------------------------------------
struct bit_reader {
  const uint32_t* img;
  const uint32_t* win;   // this lane's window row in LDS: words [base, base + win_words) of the image, MSB-first
  uint64_t words;        // words of the image that exist
  uint64_t base;         // word index of the window start
  uint64_t limit;        // give up beyond this absolute bit position
  uint32_t rp;           // bit offset of the first unread bit from the window start
  uint32_t win_words;
  bool dead;

  __device__ __forceinline__ uint64_t pos() const { return base * 32 + rp; }
  __device__ __forceinline__ uint32_t far_word(uint32_t idx) const
  {
    const uint64_t i = base + idx;
    return (i < words) ? __builtin_bswap32(img[i]) : 0u;
  }
  // the 32 bits at the read position
  __device__ __forceinline__ uint32_t peek() const
  {
    const uint32_t idx = rp >> 5, sh = rp & 31;
    uint32_t hi, lo;
    if (idx + 1 < win_words) { hi = win[idx]; lo = win[idx + 1]; }
    else { hi = far_word(idx); lo = far_word(idx + 1); }
    return sh ? ((hi << sh) | (lo >> (32 - sh))) : hi;
  }
  __device__ __forceinline__ void open(const uint32_t* image, uint64_t image_bytes, uint64_t byte_off, const uint32_t* window)
  {
    img = image; win = window; words = (image_bytes + 3) >> 2; limit = image_bytes * 8 + 64; dead = false;
    base = byte_off >> 2; rp = 8u * (uint32_t)(byte_off & 3); win_words = 0;
  }
  // n = 0..32 bits as an unsigned number
  __device__ __forceinline__ uint32_t get(uint32_t n)
  {
    if (n == 0 || dead) { return 0; }
    const uint32_t v = peek() >> (32 - n);
    rp += n;
    return v;
  }
  // any width: the low 32 bits of the number (reference reads up to 64 bits and the callers truncate)
  __device__ __forceinline__ uint32_t get_wide(uint32_t n)
  {
    while (n > 32) { (void)get(32); n -= 32; }
    return get(n);
  }
  // zeros before the next 1 bit; the 1 is consumed
  __device__ __forceinline__ uint32_t zero_run()
  {
    uint32_t run = 0;
    if (dead) { return 0; }
    for (;;) {
      const uint32_t v = peek();
      if (v != 0) {
        const uint32_t z = (uint32_t)__clz((int)v);
        rp += z + 1;
        return run + z;
      }
      run += 32; rp += 32;
      if (pos() > limit) { dead = true; return run; }
    }
  }
  __device__ __forceinline__ void align() { rp = (rp + 7u) & ~7u; }
};

// the wave moves every owner lane's window to its read position and loads the words from there on
__device__ __forceinline__ void fill_windows(bit_reader& rd, uint32_t* s_win, uint32_t row_words, uint32_t win_words,
                                             uint32_t owners, uint32_t lane)
{
  __syncthreads();
  rd.base += rd.rp >> 5; rd.rp &= 31;
  for (uint32_t o = 0; o < owners; o++) {
    const uint64_t base = ((uint64_t)(uint32_t)__shfl((int)(uint32_t)(rd.base >> 32), (int)o) << 32) | (uint32_t)__shfl((int)(uint32_t)rd.base, (int)o);
    for (uint32_t k = lane; k < win_words; k += 64) {
      const uint64_t i = base + k;
      s_win[o * row_words + k] = (i < rd.words) ? __builtin_bswap32(rd.img[i]) : 0u;
    }
  }
  rd.win_words = (lane < owners) ? win_words : 0u;
  __syncthreads();
}

__device__ __forceinline__ uint32_t gamma_get(bit_reader& rd)                  // src/SLACoder.c:140-163
{
  const uint32_t nd = rd.zero_run() + 1;
  if (nd == 1) { return 0; }
  const uint32_t top = (nd - 1 < 32) ? (1u << (nd - 1)) : 0u;
  return top + rd.get_wide(nd - 1) - 1u;
}

// ---------------------------------------------------------------------------------------------
// k_dec_crc: one wave per block, slice-parallel CRC16-IBM over [8, byte_len) (sla_crc_dev.h).
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256)
void k_dec_crc(const uint8_t* __restrict__ bytes, const sla_hip_dec_block* __restrict__ blocks, uint32_t num_blocks,
               sla_hip_dec_info* __restrict__ info)
{
  __shared__ uint16_t table[256];
  crc16_build_table(table);
  __syncthreads();
  const uint32_t lane = threadIdx.x & 63;
  const uint32_t j = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (j >= num_blocks) { return; }
  const sla_hip_dec_block b = blocks[j];
  const uint32_t crc = crc16_wave(bytes, b.byte_off + 8, b.byte_off + b.byte_len, table, lane);
  if (lane == 0) { info[j].crc = crc; }
}

// ---------------------------------------------------------------------------------------------
// k_dec_bits: one lane per block.  Parses the block header (per channel: shift, PARCOR codes, long-term
// flag / pitch / taps, initial Rice parameter) and then the body, whose codewords are interleaved
// (sample-major, channel-minor) and whose two Rice parameters per channel adapt with every sample: a
// strictly serial walk, parallel only across blocks.  `lanes` < 64 spreads the blocks over more waves
// (fewer lanes per wave = less divergence, more SIMDs and memory pipes in use); lanes * channels <= 64.
// Output: folded-back residuals (or raw samples / zeros) in the channel planes, right-justified.
// ---------------------------------------------------------------------------------------------
struct dec_bits_args {
  const uint32_t* image; uint64_t image_bytes;
  const sla_hip_dec_block* blocks; uint32_t num_blocks;
  uint32_t bps, lshift, mid_side, order, ntaps, lanes;
  int32_t* planes; uint64_t stride;
  sla_hip_dec_info* info; sla_hip_dec_chan* chan; int32_t* kint;
};

#define DEC_TILE 64          // samples per channel a lane decodes between two flushes
#define DEC_ROW  (DEC_TILE + 1)
#define DEC_WIN  128         // words of stream per channel a lane finds in LDS per tile (64 bits per sample)

template <int C>
__global__ __launch_bounds__(64)
void k_dec_bits(const dec_bits_args a)
{
  // [lanes * C] rows of DEC_TILE samples, one word of padding per row: a lane writes its rows sample by sample
  // (all lanes at the same column -> different banks), the flush reads them row-wise and stores 256 B at a time.
  // Direct 4-byte stores from the decode loop would sit in the same counter as the reader's prefetch and make
  // every refill wait for them.
  extern __shared__ __attribute__((aligned(16))) uint32_t s_dyn[];
  constexpr uint32_t WIN = DEC_WIN * C, WROW = WIN + 1;      // one word of padding: the lanes' rows start in different banks
  int32_t* s_tile = (int32_t*)s_dyn;                         // [lanes * C][DEC_ROW]
  uint32_t* s_win = s_dyn + a.lanes * C * DEC_ROW;           // [lanes][WROW]
  const uint32_t lane = threadIdx.x;
  const uint32_t j = blockIdx.x * a.lanes + lane;
  const bool mine = (lane < a.lanes) && (j < a.num_blocks);
  sla_hip_dec_block b;
  b.byte_off = 0; b.byte_len = 0; b.smp_off = 0; b.num_samples = 0; b.flags = 0;
  if (mine) { b = a.blocks[j]; }
  bit_reader rd;
  rd.open(a.image, a.image_bytes, b.byte_off, s_win + (mine ? lane : 0u) * WROW);
  fill_windows(rd, s_win, WROW, WIN, a.lanes, lane);       // helper lanes have no row of their own: they only carry words
  (void)rd.get(16);                                        // sync code          src/SLADecoder.c:330-334
  (void)rd.get(32);                                        // size field         (the host walked these)
  (void)rd.get(16);                                        // CRC16
  (void)rd.get(16);                                        // samples per channel
  const uint32_t type = mine ? rd.get(2) : 1u;
  uint32_t init[C];
#pragma unroll
  for (uint32_t c = 0; c < (uint32_t)C; c++) { init[c] = 0; }
  if (mine && type == 0) {
#pragma unroll
    for (uint32_t c = 0; c < (uint32_t)C; c++) {
      const uint32_t rsh = rd.get(4);
      int32_t* kout = a.kint + ((uint64_t)j * C + c) * (a.order + 1);
      kout[0] = 0;
      for (uint32_t ord = 1; ord <= a.order; ord++) {
        const uint32_t q = (ord < 4) ? 16u : 8u;           // SLA_GET_PARCOR_QUANTIZE_BIT_WIDTH
        const int32_t code = unfold_i32(rd.get(q));
        kout[ord] = (int32_t)((uint32_t)code << (16u - q)) >> rsh;
      }
      sla_hip_dec_chan ci;
      ci.pitch = 0;
      for (int k = 0; k < 5; k++) { ci.ltm_coef[k] = 0; }
      if (rd.get(1) != 0) {
        ci.pitch = rd.get(10);                             // SLALONGTERM_PERIOD_NUM_BITS
        for (uint32_t k = 0; k < a.ntaps; k++) {
          const int32_t q16 = unfold_i32(rd.get(16));
          if (k < 5) { ci.ltm_coef[k] = (int32_t)((uint32_t)q16 << 16); }
        }
      }
      init[c] = rd.get_wide(a.bps);
      ci.rice_init = init[c];
      ci.reserved = 0;
      a.chan[(uint64_t)j * C + c] = ci;
    }
  }
  rd.align();

  const uint32_t n = (!mine || (b.flags & SLA_HIP_DEC_HEADER_ONLY)) ? 0u : b.num_samples;
  uint64_t p0[C], p1[C];
  uint32_t gm[C];
  uint64_t avg = 0;
#pragma unroll
  for (uint32_t c = 0; c < (uint32_t)C; c++) {
    p0[c] = p1[c] = (uint64_t)(uint32_t)(init[c] << 8);                        // SLACODER_PARAMETER_SET
    uint32_t g = (uint32_t)((p0[c] + 128u) >> 8);
    gm[c] = g ? g : 1u;
    avg += gm[c];
  }
  avg /= (uint32_t)C;
  const bool adaptive = (avg > 8);                                               // SLACODER_LOW_THRESHOULD_PARAMETER
  int32_t* row = s_tile + lane * C * DEC_ROW;
  const uint32_t nmax = umax_wave(n);

  for (uint32_t s0 = 0; s0 < nmax; s0 += DEC_TILE) {
    const uint32_t cnt = (s0 < n) ? ((n - s0 < DEC_TILE) ? (n - s0) : (uint32_t)DEC_TILE) : 0u;
    fill_windows(rd, s_win, WROW, WIN, a.lanes, lane);
    if (type == 1) {
      for (uint32_t u = 0; u < cnt; u++) {
#pragma unroll
        for (uint32_t c = 0; c < (uint32_t)C; c++) { row[c * DEC_ROW + u] = 0; }
      }
    } else if (type == 2) {
      for (uint32_t u = 0; u < cnt; u++) {
#pragma unroll
        for (uint32_t c = 0; c < (uint32_t)C; c++) {
          const uint32_t nb = a.bps - a.lshift + ((c == 1 && a.mid_side) ? 1u : 0u);
          row[c * DEC_ROW + u] = unfold_i32(rd.get_wide(nb));
        }
      }
    } else if (type == 0 && adaptive) {
      for (uint32_t u = 0; u < cnt; u++) {
#pragma unroll
        for (uint32_t c = 0; c < (uint32_t)C; c++) {
          const uint32_t k0 = rice_k(p0[c]), k1 = rice_k(p1[c]);
          const uint32_t m0 = 1u << k0;
          const uint32_t v = rd.peek();
          uint32_t q = (uint32_t)__clz((int)v);                      // 32 when v == 0
          const uint32_t k = q ? k1 : k0;
          uint32_t val;
          if (q < 16u && q + 1u + k <= 32u && !rd.dead) {
            // the whole codeword (unary part, stop bit, k remainder bits) lies inside the 32 bits just read
            const uint32_t rest = k ? ((v << (q + 1u)) >> (32u - k)) : 0u;
            rd.rp += q + 1u + k;
            val = q ? (m0 + ((q - 1u) << k1) + rest) : rest;
          } else {
            q = rd.zero_run();
            if (q == 0) { val = rd.get(k0); }
            else {
              if (q == 16) { q += gamma_get(rd); }
              val = m0 + ((q - 1u) << k1) + rd.get(k1);
            }
          }
          p0[c] = rice_adapt(p0[c], val);
          if (q != 0) { p1[c] = rice_adapt(p1[c], val - m0); }
          row[c * DEC_ROW + u] = unfold_i32(val);
        }
      }
    } else if (type == 0) {
      for (uint32_t u = 0; u < cnt; u++) {
#pragma unroll
        for (uint32_t c = 0; c < (uint32_t)C; c++) {
          const uint32_t m = gm[c];
          const uint32_t q = rd.zero_run();
          uint32_t val;
          if ((m & (m - 1u)) == 0) {
            val = q * m + rd.get(ceil_log2_u32(m));
          } else {
            const uint32_t bb = ceil_log2_u32(m), cut = (1u << bb) - m;
            uint32_t rest = rd.get(bb - 1);
            if (rest < cut) { val = q * m + rest; }
            else { rest = (rest << 1) + rd.get(1); val = q * m + rest - cut; }
          }
          row[c * DEC_ROW + u] = unfold_i32(val);
        }
      }
    }
    __syncthreads();
    // flush: row r = (owner lane, channel); the 64 lanes store its 64 samples side by side
    for (uint32_t r = 0; r < a.lanes * (uint32_t)C; r++) {
      const uint32_t owner = r / (uint32_t)C, c = r - owner * (uint32_t)C;
      const uint32_t on = (uint32_t)__shfl((int)n, (int)owner), ooff = (uint32_t)__shfl((int)b.smp_off, (int)owner);
      const uint32_t s = s0 + lane;
      if (s < on) { a.planes[(uint64_t)c * a.stride + ooff + s] = s_tile[r * DEC_ROW + lane]; }
    }
    __syncthreads();
  }
  rd.align();
  if (mine) {
    sla_hip_dec_info* io = a.info + j;
    io->type = type;
    io->used_bytes = (uint32_t)((rd.pos() >> 3) - b.byte_off);
    io->overrun = (rd.dead || (rd.pos() >> 3) > a.image_bytes) ? 1u : 0u;
  }
}

// ---------------------------------------------------------------------------------------------
// k_dec_lms: sign-log LMS synthesis.  Same lane layout as the encoder's k_tail: G = 2*ORDER lanes per
// (block, channel), lane t holds coefficient t and history t (t < ORDER: past OUTPUTS, else past predictions).
// The residual e is known in advance here, so its sign and step are off the serial chain; what is serial is
// coefficient -> prediction -> output -> history.
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ int32_t sgn(int32_t v)
{
  int32_t r;
  asm("v_med3_i32 %0, %1, -1, 1" : "=v"(r) : "v"(v));
  return r;
}
__device__ __forceinline__ int32_t mad24(int32_t a, int32_t b, int32_t c)
{
  int32_t r;
  asm("v_mad_i32_i24 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
  return r;
}
template <int CTRL>
__device__ __forceinline__ uint32_t dpp_u32(uint32_t x)
{
  return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, CTRL, 0xF, 0xF, true);
}
template <int G>
__device__ __forceinline__ uint32_t group_sum(uint32_t x)
{
  x += dpp_u32<0xB1>(x);                       // quad_perm [1,0,3,2]
  x += dpp_u32<0x4E>(x);                       // quad_perm [2,3,0,1]
  x += dpp_u32<0x141>(x);                      // row_half_mirror
  if (G >= 16) { x += dpp_u32<0x140>(x); }     // row_mirror
  if (G >= 32) { x += (uint32_t)__shfl_xor((int)x, 16); }
  if (G >= 64) { x += (uint32_t)__shfl_xor((int)x, 32); }
  return x;
}

template <int ORDER, bool FIRST>
__device__ __forceinline__ int32_t lms_synth_block(int32_t e_mine, uint32_t grp_base, bool is_fir_head, bool is_iir_head,
                                                   uint32_t t, int32_t& coef, int32_t& h)
{
  constexpr int G = 2 * ORDER;
  int32_t v_mine = 0;
  int32_t es[G];
#pragma unroll
  for (int u = 0; u < G; u++) { es[u] = __shfl(e_mine, (int)(grp_base + u)); }
  __builtin_amdgcn_sched_barrier(0);
#pragma unroll
  for (int u = 0; u < G; u++) {
    const int32_t e = es[u];
    int32_t v, ph;
    if (FIRST && u < ORDER) {
      v = e; ph = e;                                                           // priming   src/SLAPredictor.c:1365-1387
    } else {
      const int32_t sh = sgn(h);
      const uint32_t sum = group_sum<G>((uint32_t)coef * (uint32_t)h) + (1u << 9);
      const int32_t p = (int32_t)sum >> 10;
      v = (int32_t)((uint32_t)e + (uint32_t)p);
      const int32_t ne = (int32_t)(0u - (uint32_t)e);
      const uint32_t mag = (uint32_t)max(e, ne);
      const int32_t lg = 32 - (int32_t)__clz((int)mag);
      const int32_t sign2 = __mul24(sgn(e), sh);
      coef = mad24(sign2, lg >> 1, coef);
      ph = p;
    }
    h = (int32_t)__builtin_amdgcn_update_dpp(h, h, 0x138 /* wave_shr:1 */, 0xF, 0xF, false);
    h = is_fir_head ? v : (is_iir_head ? ph : h);
    v_mine = (t == (uint32_t)u) ? v : v_mine;
  }
  return v_mine;
}

template <int ORDER>
__global__ __launch_bounds__(256)
void k_dec_lms(int32_t* __restrict__ planes, uint64_t stride, const sla_hip_dec_block* __restrict__ blocks,
               const sla_hip_dec_info* __restrict__ info, uint32_t num_blocks, uint32_t num_channels)
{
  constexpr int G = 2 * ORDER;
  constexpr int JPW = 64 / G;
  const uint32_t lane = threadIdx.x & 63;
  const uint32_t t = lane & (G - 1);
  const uint32_t grp_base = lane - t;
  const uint32_t wave = blockIdx.x * 4 + (threadIdx.x >> 6);
  const uint32_t j = wave * JPW + (lane / G);
  const uint32_t num_jobs = num_blocks * num_channels;
  const bool have = (j < num_jobs);
  const uint32_t bi = have ? j / num_channels : 0, ch = have ? j - bi * num_channels : 0;
  const sla_hip_dec_block b = blocks[bi];
  const bool active = have && info[bi].type == 0 && !(b.flags & SLA_HIP_DEC_HEADER_ONLY);
  uint32_t n = active ? b.num_samples : 0;
  if (n < (uint32_t)ORDER) { n = 0; }                    // fewer samples than taps: everything passes through
  int32_t* io = planes + (uint64_t)ch * stride + b.smp_off;
  const bool is_fir_head = (t == 0), is_iir_head = (t == (uint32_t)ORDER);
  const uint32_t nmax = umax_wave(n);

  int32_t coef = 0, h = 0;
  int32_t e_next = (t < n) ? io[t] : 0;
  for (uint32_t s0 = 0; s0 < nmax; s0 += G) {
    const int32_t e_mine = e_next;
    const uint32_t sn = s0 + G + t;
    e_next = (sn < n) ? io[sn] : 0;
    const int32_t v = (s0 == 0) ? lms_synth_block<ORDER, true>(e_mine, grp_base, is_fir_head, is_iir_head, t, coef, h)
                                : lms_synth_block<ORDER, false>(e_mine, grp_base, is_fir_head, is_iir_head, t, coef, h);
    const uint32_t s = s0 + t;
    if (s < n) { io[s] = v; }
  }
}

// ---------------------------------------------------------------------------------------------
// k_dec_ltm: long-term synthesis, out[s] = in[s] + ((2^30 + sum_j coef[j] * out[s - delay + j]) >> 31) for
// s >= delay = pitch + taps/2.  The nearest tap lies d = pitch - (taps-1)/2 samples back, so d consecutive
// outputs are independent: one wave per (block, channel) walks the block (held in LDS) in steps of
// min(d, 64) samples.  d <= 0 never comes out of the encoder (pitch >= 3); such a stream is walked one
// sample at a time, taps that point at or past the current sample reading not-yet-synthesised input.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(64)
void k_dec_ltm(int32_t* __restrict__ planes, uint64_t stride, const sla_hip_dec_block* __restrict__ blocks,
               const sla_hip_dec_info* __restrict__ info, const sla_hip_dec_chan* __restrict__ chan,
               uint32_t num_channels, uint32_t ntaps)
{
  extern __shared__ __attribute__((aligned(16))) int32_t s_blk[];
  const uint32_t j = blockIdx.x;
  const uint32_t bi = j / num_channels, ch = j - bi * num_channels;
  const sla_hip_dec_block b = blocks[bi];
  if (info[bi].type != 0 || (b.flags & SLA_HIP_DEC_HEADER_ONLY)) { return; }
  const sla_hip_dec_chan ci = chan[j];
  const uint32_t n = b.num_samples;
  const uint32_t delay = ci.pitch + (ntaps >> 1);
  if (ci.pitch == 0 || delay >= n) { return; }
  int32_t* io = planes + (uint64_t)ch * stride + b.smp_off;
  const uint32_t lane = threadIdx.x;
  for (uint32_t s = lane; s < n; s += 64) { s_blk[s] = io[s]; }
  __syncthreads();
  const int32_t d = (int32_t)ci.pitch - (int32_t)((ntaps - 1) >> 1);
  const uint32_t step = (d < 1) ? 1u : ((d > 64) ? 64u : (uint32_t)d);
  for (uint32_t s0 = delay; s0 < n; s0 += step) {
    const uint32_t s = s0 + lane;
    if (lane < step && s < n) {
      int64_t acc = (int64_t)1 << 30;
      for (uint32_t k = 0; k < ntaps; k++) {
        const uint32_t at = s - delay + k;
        acc += (int64_t)ci.ltm_coef[k] * (int64_t)((at < n) ? s_blk[at] : 0);
      }
      s_blk[s] = (int32_t)((uint32_t)s_blk[s] + (uint32_t)(int32_t)(acc >> 31));
    }
    __syncthreads();                                       // one wave: orders the LDS writes before the next step's reads
  }
  for (uint32_t s = delay + lane; s < n; s += 64) { io[s] = s_blk[s]; }
}

// ---------------------------------------------------------------------------------------------
// k_dec_lattice: PARCOR synthesis lattice + de-emphasis.  Per sample the reference walks the stages
// m = order..1:  f_{m-1} = f_m + R(k_m b_{m-1}[n-1]),  b_m[n] = b_{m-1}[n-1] - R(k_m f_{m-1}),  R(v) = (v + 2^14) >> 15.
// Every R(k_m b_{m-1}[n-1]) depends on the PREVIOUS sample only, so the lanes form them at once and the chain
// f_order -> f_0 becomes a prefix sum (wrapping adds are associative): position q = lane*R + r holds stage
// m = G*R - q, i.e. the first stage of the walk sits at the lowest position and the output f_0 appears on the
// last lane, which also carries the one-tap de-emphasis recurrence and stores the sample.  The new b values
// move one position up (b_m feeds stage m+1), the last position takes b_0 = f_0.
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ int32_t lat_term(int32_t k, int32_t v) { return (int32_t)((uint32_t)k * (uint32_t)v + (1u << 14)) >> 15; }

template <int CTRL, int ROWS>
__device__ __forceinline__ uint32_t dpp_rows(uint32_t x)        // 0 where the source lane does not exist / row masked off
{
  return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, CTRL, ROWS, 0xF, false);
}

template <int G>
__device__ __forceinline__ uint32_t group_prefix(uint32_t x)    // inclusive prefix sum over the G lanes of a group
{
  x += dpp_u32<0x111>(x);                      // row_shr:1
  x += dpp_u32<0x112>(x);                      // row_shr:2
  x += dpp_u32<0x114>(x);                      // row_shr:4
  x += dpp_u32<0x118>(x);                      // row_shr:8
  if (G >= 32) { x += dpp_rows<0x142, 0xA>(x); }   // row_bcast:15 into rows 1 and 3
  if (G >= 64) { x += dpp_rows<0x143, 0xC>(x); }   // row_bcast:31 into rows 2 and 3
  return x;
}

template <int G, int R>
__global__ __launch_bounds__(256)
void k_dec_lattice(int32_t* __restrict__ planes, uint64_t stride, const sla_hip_dec_block* __restrict__ blocks,
                   const sla_hip_dec_info* __restrict__ info, uint32_t num_blocks, uint32_t num_channels,
                   const int32_t* __restrict__ kint, uint32_t order, uint32_t deemphasis)
{
  constexpr int JPW = 64 / G;
  // the output lane parks its 16 samples here; the group then stores them side by side (a 4-byte store per
  // sample would share a counter with the prefetch of the next inputs and stall it)
  __shared__ int32_t s_out[(256 / G) * 16];
  const uint32_t lane = threadIdx.x & 63;
  const uint32_t t = lane & (G - 1);
  const uint32_t grp_base = lane - t;
  const uint32_t wave = blockIdx.x * 4 + (threadIdx.x >> 6);
  const uint32_t j = wave * JPW + (lane / G);
  const uint32_t num_jobs = num_blocks * num_channels;
  const bool have = (j < num_jobs);
  const uint32_t bi = have ? j / num_channels : 0, ch = have ? j - bi * num_channels : 0;
  const sla_hip_dec_block b = blocks[bi];
  const bool active = have && info[bi].type == 0 && !(b.flags & SLA_HIP_DEC_HEADER_ONLY);
  const uint32_t n = active ? b.num_samples : 0;
  int32_t* io = planes + (uint64_t)ch * stride + b.smp_off;
  int32_t* park = s_out + (threadIdx.x / G) * 16;
  const bool is_last = (t == (uint32_t)(G - 1));
  const uint32_t nmax = umax_wave(n);

  int32_t k[R], bw[R];
#pragma unroll
  for (int r = 0; r < R; r++) {
    const uint32_t m = (uint32_t)(G * R) - (t * R + r);
    k[r] = (active && m <= order) ? kint[(uint64_t)j * (order + 1) + m] : 0;
    bw[r] = 0;
  }
  int32_t yprev = 0;
  int32_t e_next = (t < n) ? io[t] : 0;
  for (uint32_t s0 = 0; s0 < nmax; s0 += G) {
    const int32_t e_mine = e_next;
    const uint32_t sn = s0 + G + t;
    e_next = (sn < n) ? io[sn] : 0;
#pragma unroll 1
    for (int u0 = 0; u0 < G; u0 += 16) {
      int32_t es[16];
#pragma unroll
      for (int u = 0; u < 16; u++) { es[u] = __shfl(e_mine, (int)(grp_base + u0 + u)); }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int u = 0; u < 16; u++) {
        uint32_t c[R];
        uint32_t run = 0;
#pragma unroll
        for (int r = 0; r < R; r++) { run += (uint32_t)lat_term(k[r], bw[r]); c[r] = run; }
        const uint32_t excl = group_prefix<G>(run) - run + (uint32_t)es[u];
        int32_t f[R], nb[R];
#pragma unroll
        for (int r = 0; r < R; r++) {
          f[r] = (int32_t)(excl + c[r]);
          nb[r] = (int32_t)((uint32_t)bw[r] - (uint32_t)lat_term(k[r], f[r]));
        }
        // b moves one position up; the group's last position takes the output
        int32_t from_next = (int32_t)__builtin_amdgcn_update_dpp(f[R - 1], nb[0], 0x130 /* wave_shl:1 */, 0xF, 0xF, false);
        from_next = is_last ? f[R - 1] : from_next;
#pragma unroll
        for (int r = 0; r < R - 1; r++) { bw[r] = nb[r + 1]; }
        bw[R - 1] = from_next;
        // de-emphasis on the output lane: y[n] = x[n] + ((y[n-1] * 31) >> 5)     src/SLAPredictor.c:1781-1786
        const int32_t y = deemphasis ? (int32_t)((uint32_t)f[R - 1] + (uint32_t)((int32_t)((uint32_t)yprev * 31u) >> 5)) : f[R - 1];
        yprev = y;
        if (is_last) { park[u] = y; }
      }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
      {
        const uint32_t s = s0 + (uint32_t)u0 + (t & 15u);
        if (t < 16u && s < n) { io[s] = park[t & 15u]; }
      }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    }
  }
}

// ---------------------------------------------------------------------------------------------
// k_dec_finish: mid/side -> left/right and the final left-justification, elementwise over all samples.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256)
void k_dec_finish(int32_t* __restrict__ planes, uint64_t stride, uint32_t num_channels, uint32_t num_samples,
                  uint32_t mid_side, uint32_t shift)
{
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= num_samples) { return; }
  if (mid_side) {
    const int32_t side = planes[stride + i];
    const int32_t mid = (int32_t)(((uint32_t)planes[i] << 1) | ((uint32_t)side & 1u));
    const int32_t l = (int32_t)((uint32_t)mid + (uint32_t)side) >> 1, r = (int32_t)((uint32_t)mid - (uint32_t)side) >> 1;
    planes[i] = (int32_t)((uint32_t)l << shift);
    planes[stride + i] = (int32_t)((uint32_t)r << shift);
    for (uint32_t c = 2; c < num_channels; c++) { planes[(uint64_t)c * stride + i] = (int32_t)((uint32_t)planes[(uint64_t)c * stride + i] << shift); }
  } else {
    for (uint32_t c = 0; c < num_channels; c++) { planes[(uint64_t)c * stride + i] = (int32_t)((uint32_t)planes[(uint64_t)c * stride + i] << shift); }
  }
}

// De-emphasis as a pass of its own (per-call API): y[n] = x[n] + ((y[n-1] * (2^s - 1)) >> s), y[-1] = previous.
// A one-tap recurrence through a truncating shift: strictly serial, one lane.
__global__ __launch_bounds__(64)
void k_dec_deemphasis(int32_t* __restrict__ data, uint32_t n, int32_t previous, uint32_t shift)
{
  if (threadIdx.x != 0 || blockIdx.x != 0) { return; }
  const uint32_t numer = (1u << shift) - 1u;
  int32_t y = previous;
  for (uint32_t i = 0; i < n; i++) {
    y = (int32_t)((uint32_t)data[i] + (uint32_t)((int32_t)((uint32_t)y * numer) >> shift));
    data[i] = y;
  }
}

}  // namespace

// ---------------------------------------------------------------------------------------------
// launchers
// ---------------------------------------------------------------------------------------------
extern "C" int sla_hip_launch_dec_bits(const uint32_t* d_image, uint64_t image_bytes,
                                       const sla_hip_dec_block* d_blocks, uint32_t num_blocks,
                                       uint32_t num_channels, uint32_t bits_per_sample, uint32_t offset_lshift,
                                       uint32_t mid_side, uint32_t parcor_order, uint32_t longterm_order,
                                       uint32_t want_crc, int32_t* d_planes, uint64_t plane_stride,
                                       sla_hip_dec_info* d_info, sla_hip_dec_chan* d_chan, int32_t* d_kint,
                                       sla_hip_stream_t stream)
{
  if (d_image == nullptr || d_blocks == nullptr || d_planes == nullptr || d_info == nullptr || d_chan == nullptr || d_kint == nullptr) {
    return SLA_APIRESULT_INVALID_ARGUMENT;
  }
  if (num_channels == 0 || num_channels > 8 || bits_per_sample == 0 || bits_per_sample > 32 || offset_lshift >= bits_per_sample
      || parcor_order > 255 || longterm_order > 5 || (mid_side && num_channels != 2)) { return SLA_APIRESULT_INVALID_ARGUMENT; }
  if (num_blocks == 0) { return 0; }
  hipStream_t st = (hipStream_t)stream;
  if (want_crc) {
    hipLaunchKernelGGL(k_dec_crc, dim3((num_blocks + 3) / 4), dim3(256), 0, st, (const uint8_t*)d_image, d_blocks, num_blocks, d_info);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { return hip_rc(e); }
  }
  dec_bits_args a;
  a.image = d_image; a.image_bytes = image_bytes; a.blocks = d_blocks; a.num_blocks = num_blocks;
  a.bps = bits_per_sample; a.lshift = offset_lshift; a.mid_side = mid_side; a.order = parcor_order; a.ntaps = longterm_order;
  // enough waves for every SIMD of the chip before a wave takes a second block
  uint32_t lanes = (num_blocks + 1023) / 1024;
  if (lanes < 1) { lanes = 1; }
  if (lanes > 64 / num_channels) { lanes = 64 / num_channels; }     // lanes * channels rows of LDS staging
  a.lanes = lanes;
  a.planes = d_planes; a.stride = plane_stride; a.info = d_info; a.chan = d_chan; a.kint = d_kint;
  const dim3 grid((num_blocks + lanes - 1) / lanes), block(64);
  const size_t lds = sizeof(uint32_t) * ((size_t)lanes * num_channels * DEC_ROW + (size_t)lanes * (DEC_WIN * num_channels + 1));
  switch (num_channels) {
    case 1: hipLaunchKernelGGL(k_dec_bits<1>, grid, block, lds, st, a); break;
    case 2: hipLaunchKernelGGL(k_dec_bits<2>, grid, block, lds, st, a); break;
    case 3: hipLaunchKernelGGL(k_dec_bits<3>, grid, block, lds, st, a); break;
    case 4: hipLaunchKernelGGL(k_dec_bits<4>, grid, block, lds, st, a); break;
    case 5: hipLaunchKernelGGL(k_dec_bits<5>, grid, block, lds, st, a); break;
    case 6: hipLaunchKernelGGL(k_dec_bits<6>, grid, block, lds, st, a); break;
    case 7: hipLaunchKernelGGL(k_dec_bits<7>, grid, block, lds, st, a); break;
    default: hipLaunchKernelGGL(k_dec_bits<8>, grid, block, lds, st, a); break;
  }
  return hip_rc(hipGetLastError());
}

extern "C" int sla_hip_launch_dec_lms(int32_t* d_planes, uint64_t plane_stride, const sla_hip_dec_block* d_blocks,
                                      const sla_hip_dec_info* d_info, uint32_t num_blocks, uint32_t num_channels,
                                      uint32_t lms_order, sla_hip_stream_t stream)
{
  if (d_planes == nullptr || d_blocks == nullptr || d_info == nullptr) { return SLA_APIRESULT_INVALID_ARGUMENT; }
  if (num_channels == 0 || num_channels > 8) { return SLA_APIRESULT_INVALID_ARGUMENT; }
  if (lms_order != 4 && lms_order != 8 && lms_order != 16 && lms_order != 32) { return SLA_APIRESULT_INVALID_ARGUMENT; }
  if (num_blocks == 0) { return 0; }
  hipStream_t st = (hipStream_t)stream;
  const uint32_t jobs = num_blocks * num_channels, jpw = 64 / (2 * lms_order);
  const dim3 grid(((jobs + jpw - 1) / jpw + 3) / 4), block(256);
  switch (lms_order) {
    case 4:  hipLaunchKernelGGL(k_dec_lms<4>,  grid, block, 0, st, d_planes, plane_stride, d_blocks, d_info, num_blocks, num_channels); break;
    case 8:  hipLaunchKernelGGL(k_dec_lms<8>,  grid, block, 0, st, d_planes, plane_stride, d_blocks, d_info, num_blocks, num_channels); break;
    case 16: hipLaunchKernelGGL(k_dec_lms<16>, grid, block, 0, st, d_planes, plane_stride, d_blocks, d_info, num_blocks, num_channels); break;
    default: hipLaunchKernelGGL(k_dec_lms<32>, grid, block, 0, st, d_planes, plane_stride, d_blocks, d_info, num_blocks, num_channels); break;
  }
  return hip_rc(hipGetLastError());
}

extern "C" int sla_hip_launch_dec_ltm(int32_t* d_planes, uint64_t plane_stride, const sla_hip_dec_block* d_blocks,
                                      const sla_hip_dec_info* d_info, const sla_hip_dec_chan* d_chan,
                                      uint32_t num_blocks, uint32_t num_channels, uint32_t longterm_order,
                                      uint32_t max_block_samples, sla_hip_stream_t stream)
{
  if (d_planes == nullptr || d_blocks == nullptr || d_info == nullptr || d_chan == nullptr) { return SLA_APIRESULT_INVALID_ARGUMENT; }
  if (num_channels == 0 || num_channels > 8 || longterm_order > 5) { return SLA_APIRESULT_INVALID_ARGUMENT; }
  if ((size_t)max_block_samples * sizeof(int32_t) > SLA_HIP_LDS_BUDGET) { return SLA_APIRESULT_EXCEED_HANDLE_CAPACITY; }
  if (num_blocks == 0 || longterm_order == 0) { return 0; }
  const size_t lds = (size_t)max_block_samples * sizeof(int32_t);
  if (lds > 64 * 1024) {
    hipError_t e = hipFuncSetAttribute((const void*)k_dec_ltm, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) { return hip_rc(e); }
  }
  hipLaunchKernelGGL(k_dec_ltm, dim3(num_blocks * num_channels), dim3(64), lds, (hipStream_t)stream, d_planes, plane_stride,
                     d_blocks, d_info, d_chan, num_channels, longterm_order);
  return hip_rc(hipGetLastError());
}

extern "C" int sla_hip_launch_dec_lattice(int32_t* d_planes, uint64_t plane_stride, const sla_hip_dec_block* d_blocks,
                                          const sla_hip_dec_info* d_info, uint32_t num_blocks, uint32_t num_channels,
                                          const int32_t* d_kint, uint32_t parcor_order, uint32_t deemphasis, sla_hip_stream_t stream)
{
  if (d_planes == nullptr || d_blocks == nullptr || d_info == nullptr || d_kint == nullptr) { return SLA_APIRESULT_INVALID_ARGUMENT; }
  if (num_channels == 0 || num_channels > 8 || parcor_order > 255) { return SLA_APIRESULT_INVALID_ARGUMENT; }
  if (num_blocks == 0) { return 0; }
  hipStream_t st = (hipStream_t)stream;
  const uint32_t jobs = num_blocks * num_channels;
#define SLA_DEC_LATTICE(G, R)                                                                                          \
  hipLaunchKernelGGL((k_dec_lattice<G, R>), dim3(((jobs + (64 / G) - 1) / (64 / G) + 3) / 4), dim3(256), 0, st, d_planes, \
                     plane_stride, d_blocks, d_info, num_blocks, num_channels, d_kint, parcor_order, deemphasis)
  if (parcor_order <= 16) { SLA_DEC_LATTICE(16, 1); }
  else if (parcor_order <= 32) { SLA_DEC_LATTICE(32, 1); }
  else if (parcor_order <= 64) { SLA_DEC_LATTICE(64, 1); }
  else if (parcor_order <= 128) { SLA_DEC_LATTICE(64, 2); }
  else { SLA_DEC_LATTICE(64, 4); }
#undef SLA_DEC_LATTICE
  return hip_rc(hipGetLastError());
}

extern "C" int sla_hip_launch_dec_finish(int32_t* d_planes, uint64_t plane_stride, uint32_t num_channels,
                                         uint32_t num_samples, uint32_t mid_side, uint32_t shift, sla_hip_stream_t stream)
{
  if (d_planes == nullptr) { return SLA_APIRESULT_INVALID_ARGUMENT; }
  if (num_channels == 0 || num_channels > 8 || shift > 31 || (mid_side && num_channels != 2) || plane_stride < num_samples) {
    return SLA_APIRESULT_INVALID_ARGUMENT;
  }
  if (num_samples == 0) { return 0; }
  hipLaunchKernelGGL(k_dec_finish, dim3((num_samples + 255) / 256), dim3(256), 0, (hipStream_t)stream, d_planes, plane_stride,
                     num_channels, num_samples, mid_side, shift);
  return hip_rc(hipGetLastError());
}

extern "C" int sla_hip_launch_dec_deemphasis(int32_t* d_data, uint32_t num_samples, int32_t previous, uint32_t coef_shift,
                                             sla_hip_stream_t stream)
{
  if (d_data == nullptr || coef_shift < 1 || coef_shift > 30) { return SLA_APIRESULT_INVALID_ARGUMENT; }
  if (num_samples == 0) { return 0; }
  hipLaunchKernelGGL(k_dec_deemphasis, dim3(1), dim3(64), 0, (hipStream_t)stream, d_data, num_samples, previous, coef_shift);
  return hip_rc(hipGetLastError());
}
