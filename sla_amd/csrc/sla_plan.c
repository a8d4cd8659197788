/*
 * sla_plan.c -- host-side scalar decisions of the encode path.
 *
 * Everything here consumes a handful of doubles per block that the kernels
 * produced (r[0], PARCOR) and must use the host libm so that the partition
 * choice and the RAW fallback compare exactly as the reference does:
 * transcendental functions are never evaluated on the device.
 */
#include "sla_internal.h"

#include <float.h>
#include <math.h>

#define SLAI_PI 3.1415926535897932384626433832795029

/* analysis windows (reference src/SLAUtility.c:99-189) */
int slai_make_window(SLAWindowFunctionType type, double* w, uint32_t n)
{
  uint32_t i;
  if ((int)type < 0 || (int)type > (int)SLA_WINDOWFUNCTIONTYPE_VORBIS) { return -1; }
  if (type == SLA_WINDOWFUNCTIONTYPE_RECTANGULAR) { for (i = 0; i < n; i++) { w[i] = 1.0; } return 0; }
  if (n == 1) { w[0] = 1.0; return 0; }
  for (i = 0; i < n; i++) {
    const double x = (double)i / (n - 1);
    switch (type) {
      case SLA_WINDOWFUNCTIONTYPE_SIN:      w[i] = sin(SLAI_PI * x); break;
      case SLA_WINDOWFUNCTIONTYPE_HANN:     w[i] = 0.5f - 0.5f * cos(2.0f * SLAI_PI * x); break;
      case SLA_WINDOWFUNCTIONTYPE_BLACKMAN: w[i] = 0.42f - 0.5f * cos(2.0f * SLAI_PI * x) + 0.08f * cos(4.0f * SLAI_PI * x); break;
      default:                              w[i] = sin((SLAI_PI / 2.0f) * sin(SLAI_PI * x) * sin(SLAI_PI * x)); break;
    }
  }
  return 0;
}

static double log2_libm(double x) { return log(x) * 1.4426950408889634; }   /* src/SLAUtility.c:442-447 */

/* bytes per sample from the block's energy and reflection coefficients
 * (reference src/SLAPredictor.c:416-468; sumsq is the device's r[0]) */
double slai_code_length(double sumsq, uint32_t n, uint32_t bps, const double* parcor, uint32_t order)
{
  double power = sumsq * ldexp(1.0, (int)(2 * (bps - 1)));
  double gain = 0.0, len;
  uint32_t ord;
  if (fabs(power) <= FLT_MIN) { return 0.0; }
  power = log2_libm(power) - log2_libm((double)n);
  for (ord = 1; ord <= order; ord++) { gain += log2_libm(1.0 - parcor[ord] * parcor[ord]); }
  len = 1.9426950408889634 + 0.5f * (power + gain);
  len /= 8;
  return (len <= 0) ? (1.0f / 8) : len;
}

/* Dijkstra from node 0 to node nodes-1 over a dense matrix, first-minimum
 * selection and strict-improvement relaxation as the reference
 * (src/SLAPredictor.c:1521-1581).  path[i] = predecessor.  Returns 0, or -1
 * when the goal cannot be settled (the reference would not terminate). */
int slai_shortest_path(const double* adj, uint32_t nodes, uint32_t* path)
{
  double cost[SLAI_MAX_NODES];
  uint8_t done[SLAI_MAX_NODES];
  uint32_t i, cur = 0, round;
  if (nodes < 2 || nodes > SLAI_MAX_NODES) { return -1; }
  for (i = 0; i < nodes; i++) { cost[i] = SLAI_BIG_WEIGHT; done[i] = 0; path[i] = 0xFFFFFFFFu; }
  cost[0] = 0.0;
  for (round = 0; round <= nodes; round++) {
    double best = SLAI_BIG_WEIGHT;
    for (i = 0; i < nodes; i++) { if (!done[i] && cost[i] < best) { best = cost[i]; cur = i; } }
    if (cur == nodes - 1) { return 0; }
    for (i = 0; i < nodes; i++) {
      const double via = adj[cur * nodes + i] + cost[cur];
      if (cost[i] > via) { cost[i] = via; path[i] = cur; }
    }
    done[cur] = 1;
  }
  return -1;
}

/* number of consecutive all-zero samples starting at `from`, capped at `limit`
 * (the bit mask has one bit per sample, set when any channel is non-zero) */
uint32_t slai_zero_run(const uint64_t* nz, uint64_t from, uint64_t limit)
{
  uint64_t pos = from, end = from + limit;
  if (nz == NULL) { return 0; }                 /* no mask: nothing is silent */
  while (pos < end) {
    uint64_t word = nz[pos >> 6] >> (pos & 63);
    if (word != 0) {
      pos += (uint64_t)__builtin_ctzll(word);
      break;
    }
    pos += 64 - (pos & 63);
  }
  if (pos > end) { pos = end; }
  return (uint32_t)(pos - from);
}

int slai_range_is_zero(const uint64_t* nz, uint64_t from, uint64_t count)
{
  return slai_zero_run(nz, from, count) == count;
}

/* Bit-identical output leans on two host properties the reference's own x86-64 build has: the Toeplitz solve
 * refines its residual in x87 extended precision (long double with a 64-bit significand, src/SLAUtility.c:627-657),
 * and windows / code lengths come from the C library's sin(), cos(), log().  The second cannot be checked against
 * fixed answers (the library may pick a different kernel per CPU and still be what the reference would get on this
 * very host), the first can: 0 = this host evaluates long double the way the reference build does. */
#include <float.h>
int slai_host_check(void)
{
#if !defined(__x86_64__) || LDBL_MANT_DIG != 64
  return 1;
#else
  volatile long double one = 1.0L, tiny = 0x1p-63L;
  volatile long double sum = one + tiny;                 /* representable only with a 64-bit significand */
  volatile double d = 1.0, dt = 0x1p-53;
  volatile double dsum = d + dt;                         /* and plain doubles must round to 53 bits (no x87 double rounding) */
  return (sum != one && dsum == d) ? 0 : 1;
#endif
}
