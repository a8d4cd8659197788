/*
 * sla_pack.c -- the bit-serial tail that stays on the host (north-star design):
 * block header + coefficient fields (reference src/SLAEncoder.c:682-737), RAW
 * body (:741-764), recursive-Rice / Golomb body (src/SLACoder.c:45-82,
 * 120-138, 224-270, 429-467), CRC16-IBM (src/SLAUtility.c:322-339) and the
 * 43-byte file header (src/SLAEncoder.c:227-292).
 *
 * Blocks are independent byte strings, so the caller packs them on several
 * host threads and concatenates.
 */
#include "sla_internal.h"

#include <string.h>

/* ---- CRC16-IBM, reflected polynomial 0xA001, init 0 ----------------------- */
static uint16_t g_crc[256];
static volatile int g_crc_ready = 0;
static void crc_setup(void)
{
  uint32_t i, k;
  for (i = 0; i < 256; i++) {
    uint32_t c = i;
    for (k = 0; k < 8; k++) { c = (c & 1u) ? ((c >> 1) ^ 0xA001u) : (c >> 1); }
    g_crc[i] = (uint16_t)c;
  }
  g_crc_ready = 1;
}
uint32_t slai_crc16(const uint8_t* data, size_t n)
{
  uint32_t crc = 0;
  size_t i;
  if (!g_crc_ready) { crc_setup(); }
  for (i = 0; i < n; i++) { crc = (crc >> 8) ^ g_crc[(crc ^ data[i]) & 0xFFu]; }
  return crc & 0xFFFFu;
}

/* ---- MSB-first bit writer with a 64-bit accumulator ------------------------ */
typedef struct { uint8_t* p; uint8_t* end; uint64_t acc; uint32_t fill; int overflow; } bits_t;

static inline void bits_flush_bytes(bits_t* b)
{
  while (b->fill >= 8) {
    b->fill -= 8;
    if (b->p < b->end) { *b->p++ = (uint8_t)(b->acc >> b->fill); } else { b->overflow = 1; }
  }
}
static inline void bits_put(bits_t* b, uint32_t v, uint32_t n)   /* 1 <= n <= 32 */
{
  const uint64_t m = (n >= 32) ? 0xFFFFFFFFull : ((1ull << n) - 1ull);
  b->acc = (b->acc << n) | ((uint64_t)v & m);
  b->fill += n;
  bits_flush_bytes(b);
}
static inline void bits_zeros(bits_t* b, uint32_t n)
{
  while (n >= 32) { bits_put(b, 0, 32); n -= 32; }
  if (n) { bits_put(b, 0, n); }
}
static inline void bits_align(bits_t* b) { if (b->fill) { bits_put(b, 0, 8 - b->fill); } }

/* ---- integer helpers -------------------------------------------------------- */
static inline uint32_t ceil_log2(uint32_t x) { return (x > 1) ? (32u - (uint32_t)__builtin_clz(x - 1u)) : 0u; }
static inline uint32_t fold(int32_t s) { const uint32_t u = (uint32_t)s << 1; return (s < 0) ? ~u : u; }

/* ---- adaptive Rice parameters: 8 fractional bits, EMA 119/128 -------------- */
typedef uint64_t rparam_t;
static inline uint32_t rp_round(rparam_t f) { return (uint32_t)((f + 128u) >> 8); }
static inline uint32_t rp_value(rparam_t f) { const uint32_t v = rp_round(f); return v ? v : 1u; }
static inline uint32_t rp_modulus(rparam_t f)   /* power of two >= mean/2 */
{
  const uint32_t v = rp_round(f >> 1);
  return 1u << ceil_log2(v ? v : 1u);
}
static inline void rp_adapt(rparam_t* f, uint32_t code)
{
  *f = (119u * (*f) + (uint64_t)(uint32_t)(9u * (uint32_t)(code << 8)) + 64u) >> 7;
}

static inline void put_unary(bits_t* b, uint32_t q) { bits_zeros(b, q); bits_put(b, 1, 1); }

static inline void put_gamma(bits_t* b, uint32_t v)
{
  if (v == 0) { bits_put(b, 1, 1); return; }
  {
    const uint32_t nd = ceil_log2(v + 2);
    bits_put(b, 0, nd - 1);
    bits_put(b, v + 1, nd);
  }
}

static inline void put_golomb(bits_t* b, uint32_t m, uint32_t v)
{
  const uint32_t q = v / m, rest = v % m;
  put_unary(b, q);
  if ((m & (m - 1)) == 0) {
    if (m > 1) { bits_put(b, rest, ceil_log2(m)); }
  } else {
    const uint32_t nb = ceil_log2(m), cut = (1u << nb) - m;
    if (rest < cut) { bits_put(b, rest, nb - 1); } else { bits_put(b, rest + cut, nb); }
  }
}

static inline void put_recursive_rice(bits_t* b, rparam_t* prm, uint32_t v)
{
  /* two parameters: stage 0 catches v < m0 with a single 1-bit prefix */
  uint32_t m = rp_modulus(prm[0]);
  if (v < m) {
    bits_put(b, 1, 1);
    if (m != 1) { bits_put(b, v & (m - 1), ceil_log2(m)); }
    rp_adapt(&prm[0], v);
    return;
  }
  rp_adapt(&prm[0], v);
  v -= m;
  m = rp_modulus(prm[1]);
  {
    const uint32_t q = 1u + v / m;
    if (q < SLAI_QUOT_THRESHOLD) { put_unary(b, q); }
    else { put_unary(b, SLAI_QUOT_THRESHOLD); put_gamma(b, q - SLAI_QUOT_THRESHOLD); }
    if (m != 1) { bits_put(b, v & (m - 1), ceil_log2(m)); }
    rp_adapt(&prm[1], v);
  }
}

/* ---- one block --------------------------------------------------------------- */
/* sync, size/crc placeholders, sample count, type, per-channel coefficient fields; byte aligned */
static void put_block_header(bits_t* b, const slai_block_params* bp)
{
  const uint32_t C = bp->num_channels, O1 = bp->order + 1;
  uint32_t ch, ord;
  bits_put(b, SLAI_SYNC_CODE, 16);
  bits_put(b, 0, 32);                        /* size, patched later */
  bits_put(b, 0, 16);                        /* crc,  patched later */
  bits_put(b, bp->num_samples, 16);
  bits_put(b, bp->type, 2);
  if (bp->type == SLAI_BLK_COMPRESS) {
    for (ch = 0; ch < C; ch++) {
      bits_put(b, bp->rshift[ch], 4);
      for (ord = 1; ord < O1; ord++) { bits_put(b, fold(bp->code[ch * O1 + ord]), (ord < 4) ? 16 : 8); }
      if (bp->pitch[ch] >= SLAI_LTM_MIN_PITCH) {
        bits_put(b, 1, 1);
        bits_put(b, bp->pitch[ch], SLAI_LTM_PERIOD_BITS);
        for (ord = 0; ord < bp->ntaps; ord++) { bits_put(b, fold(bp->ltm_q[ch * SLAI_MAX_TAPS + ord] >> 16), 16); }
      } else {
        bits_put(b, 0, 1);
      }
      bits_put(b, rp_value((rparam_t)(uint32_t)(bp->rice_init[ch] << 8)), bp->bps);
    }
  }
  bits_align(b);
}

/* header bytes only (the device packer appends the body); returns bytes written or 0 */
uint32_t slai_pack_header(const slai_block_params* bp, uint8_t* out, uint32_t cap)
{
  bits_t b;
  if (cap < 16) { return 0; }
  b.p = out; b.end = out + cap; b.acc = 0; b.fill = 0; b.overflow = 0;
  put_block_header(&b, bp);
  return b.overflow ? 0 : (uint32_t)(b.p - out);
}

/* Golomb modulus per channel when the block is coded with fixed parameters, 0 for adaptive mode
 * (mean of the channels' initial parameters <= 8, reference src/SLACoder.c:443-466) */
void slai_coding_mode(const uint32_t* rice_init, uint32_t num_channels, uint32_t* golomb_m)
{
  uint64_t mean = 0;
  uint32_t ch;
  for (ch = 0; ch < num_channels; ch++) {
    golomb_m[ch] = rp_value((rparam_t)(uint32_t)(rice_init[ch] << 8));
    mean += golomb_m[ch];
  }
  mean /= num_channels;
  if (mean > SLAI_RICE_LOW_THRESHOLD) { for (ch = 0; ch < num_channels; ch++) { golomb_m[ch] = 0; } }
}

uint32_t slai_pack_block(const slai_block_params* bp, uint8_t* out, uint32_t cap)
{
  const uint32_t C = bp->num_channels, n = bp->num_samples;
  uint32_t ch, s, size;
  bits_t b;
  if (cap < 16) { return 0; }
  b.p = out; b.end = out + cap; b.acc = 0; b.fill = 0; b.overflow = 0;
  put_block_header(&b, bp);

  if (bp->type == SLAI_BLK_RAW) {
    uint32_t width[SLAI_MAX_CHANNELS];
    for (ch = 0; ch < C; ch++) { width[ch] = bp->bps - bp->lshift + ((ch == 1 && bp->mid_side) ? 1u : 0u); }
    for (s = 0; s < n; s++) { for (ch = 0; ch < C; ch++) { bits_put(&b, fold(bp->res[ch][s]), width[ch]); } }
  } else if (bp->type == SLAI_BLK_COMPRESS) {
    rparam_t prm[SLAI_MAX_CHANNELS][SLAI_RICE_PARAMS];
    uint32_t fixed[SLAI_MAX_CHANNELS];
    uint64_t mean = 0;
    for (ch = 0; ch < C; ch++) {
      prm[ch][0] = prm[ch][1] = (rparam_t)(uint32_t)(bp->rice_init[ch] << 8);
      fixed[ch] = rp_value(prm[ch][0]);
      mean += fixed[ch];
    }
    mean /= C;
    if (mean > SLAI_RICE_LOW_THRESHOLD) {
      if (C == 1) {
        const int32_t* r0 = bp->res[0];
        for (s = 0; s < n; s++) { put_recursive_rice(&b, prm[0], fold(r0[s])); }
      } else {
        for (s = 0; s < n; s++) { for (ch = 0; ch < C; ch++) { put_recursive_rice(&b, prm[ch], fold(bp->res[ch][s])); } }
      }
    } else {
      for (s = 0; s < n; s++) { for (ch = 0; ch < C; ch++) { put_golomb(&b, fixed[ch], fold(bp->res[ch][s])); } }
    }
  }
  bits_align(&b);
  if (b.overflow) { return 0; }
  size = (uint32_t)(b.p - out);
  out[2] = (uint8_t)((size - 6) >> 24); out[3] = (uint8_t)((size - 6) >> 16);
  out[4] = (uint8_t)((size - 6) >> 8);  out[5] = (uint8_t)(size - 6);
  {
    const uint32_t crc = slai_crc16(out + SLAI_BLK_CRC_START, size - SLAI_BLK_CRC_START);
    out[6] = (uint8_t)(crc >> 8); out[7] = (uint8_t)crc;
  }
  return size;
}

/* ---- file header ---------------------------------------------------------------- */
static void be16(uint8_t* d, uint32_t v) { d[0] = (uint8_t)(v >> 8); d[1] = (uint8_t)v; }
static void be32(uint8_t* d, uint32_t v) { d[0] = (uint8_t)(v >> 24); d[1] = (uint8_t)(v >> 16); d[2] = (uint8_t)(v >> 8); d[3] = (uint8_t)v; }

int slai_write_header(const struct SLAHeaderInfo* h, uint8_t* d, uint32_t data_size)
{
  if (h == NULL || d == NULL) { return SLA_APIRESULT_INVALID_ARGUMENT; }
  if (data_size < SLA_HEADER_SIZE) { return SLA_APIRESULT_INSUFFICIENT_BUFFER_SIZE; }
  d[0] = 'S'; d[1] = 'L'; d[2] = '*'; d[3] = 1;
  be32(d + 4, SLA_HEADER_SIZE - 8);
  be16(d + 8, 0);
  be32(d + 10, SLA_FORMAT_VERSION);
  d[14] = (uint8_t)h->wave_format.num_channels;
  be32(d + 15, h->num_samples);
  be32(d + 19, h->wave_format.sampling_rate);
  d[23] = (uint8_t)h->wave_format.bit_per_sample;
  d[24] = h->wave_format.offset_lshift;
  d[25] = (uint8_t)h->encode_param.parcor_order;
  d[26] = (uint8_t)h->encode_param.longterm_order;
  d[27] = (uint8_t)h->encode_param.lms_order_per_filter;
  d[28] = (uint8_t)h->encode_param.ch_process_method;
  be32(d + 29, h->num_blocks);
  be16(d + 33, h->encode_param.max_num_block_samples);
  be32(d + 35, h->max_block_size);
  be32(d + 39, h->max_bit_per_second);
  be16(d + 8, slai_crc16(d + SLAI_HDR_CRC_START, SLA_HEADER_SIZE - SLAI_HDR_CRC_START));
  return SLA_APIRESULT_OK;
}
