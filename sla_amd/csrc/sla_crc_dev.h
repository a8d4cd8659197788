// sla_crc_dev.h -- CRC16-IBM (reflected polynomial 0xA001, start value 0, no final xor: reference
// src/SLAUtility.c:321-339) of a byte range by the 64 lanes of one wave.
//
// The CRC is linear, so the range is cut into 64 slices, every lane walks its slice through the byte table
// (start value 0), and the slice results are combined: crc(A || B) = shift(crc(A), |B|) ^ crc(B), where
// shift(c, n) = "c after n more zero bytes" = c * x^(8n) mod P in the bit-reversed (normal polynomial) domain.
// Lane 0 takes the odd bytes, all other slices are L bytes long, so lane i is followed by exactly (63 - i) * L
// bytes and the combination is a 6-level tree whose level k multiplies by g^(2^k), g = x^(8L) -- about 40 16-bit
// polynomial products per wave instead of a serial walk over the whole block.
#ifndef SLA_CRC_DEV_H_INCLUDED
#define SLA_CRC_DEV_H_INCLUDED

#include <hip/hip_runtime.h>
#include <stdint.h>

__device__ __forceinline__ void crc16_build_table(uint16_t* table /* 256 entries in LDS */)
{
  for (uint32_t i = threadIdx.x; i < 256; i += blockDim.x) {
    uint32_t c = i;
    for (int k = 0; k < 8; k++) { c = (c & 1u) ? ((c >> 1) ^ 0xA001u) : (c >> 1); }
    table[i] = (uint16_t)c;
  }
}

// a * b mod (x^16 + x^15 + x^2 + 1), 16-bit polynomials in normal bit order
__device__ __forceinline__ uint32_t crc16_mulmod(uint32_t a, uint32_t b)
{
  uint32_t r = 0;
#pragma unroll
  for (int i = 15; i >= 0; i--) {
    r = ((r << 1) ^ ((r & 0x8000u) ? 0x8005u : 0u)) & 0xFFFFu;
    r ^= ((b >> i) & 1u) ? a : 0u;
  }
  return r;
}

__device__ __forceinline__ uint32_t crc16_rev(uint32_t v) { return __brev(v) >> 16; }

// all 64 lanes of the wave call this with the same arguments; every lane returns the CRC of bytes[begin, end)
__device__ __forceinline__ uint32_t crc16_wave(const uint8_t* __restrict__ bytes, uint64_t begin, uint64_t end,
                                               const uint16_t* table, uint32_t lane)
{
  const uint64_t n = (end > begin) ? end - begin : 0;
  const uint64_t L = n / 64;                                   // lanes 1..63: L bytes; lane 0: the first n - 63 L bytes
  const uint64_t head = n - 63 * L;
  uint64_t at = begin + ((lane == 0) ? 0 : head + (uint64_t)(lane - 1) * L);
  const uint64_t stop = begin + head + (uint64_t)lane * L;
  uint32_t crc = 0;
  for (; at + 16 <= stop; at += 16) {                          // byte loads first, then the serial table walk
    uint8_t v[16];
#pragma unroll
    for (int u = 0; u < 16; u++) { v[u] = bytes[at + u]; }
#pragma unroll
    for (int u = 0; u < 16; u++) { crc = (crc >> 8) ^ table[(crc ^ v[u]) & 0xFFu]; }
  }
  for (; at < stop; at++) { crc = (crc >> 8) ^ table[(crc ^ bytes[at]) & 0xFFu]; }
  // g = x^(8L) mod P (the same in every lane: no divergence)
  uint32_t g = 1, base = 2;
  for (uint64_t k = 8 * L; k != 0; k >>= 1) {
    if (k & 1) { g = crc16_mulmod(g, base); }
    base = crc16_mulmod(base, base);
  }
  // tree: after level k the lanes whose low k+1 index bits are all ones hold the CRC of their 2^(k+1) slices
  uint32_t r = crc16_rev(crc);
#pragma unroll
  for (int k = 0; k < 6; k++) {
    const uint32_t left = (uint32_t)__shfl_up((int)r, 1 << k);         // the group of 2^k slices in front of mine
    const uint32_t joined = crc16_mulmod(left, g) ^ r;
    const bool take = (((lane >> k) & 1u) != 0) && (lane >= (1u << k));
    r = take ? joined : r;
    g = crc16_mulmod(g, g);
  }
  return crc16_rev((uint32_t)__shfl((int)r, 63));
}

#endif
