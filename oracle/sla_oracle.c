/*
 * sla_oracle.c -- CPU restatement of the SLA codec's arithmetic (oracle).
 *
 * TEST INFRASTRUCTURE ONLY: the parity checker for the HIP path in sla_amd/.
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * load libsla_oracle.so; the product never links, calls or falls back to it.
 *
 * This is a from-scratch restatement, written against the behaviour of the
 * reference (aikiriao/SLA, /root/reference); every function cites the
 * reference file:line whose arithmetic it follows.  Parity status: PINNED --
 * tests/test_oracle_vs_ref.py compares every function below (and whole-file
 * .sla bytes) with the unmodified reference compiled into oracle/_ref in the
 * build container, and tests/test_oracle_golden.py re-checks the committed
 * vectors of tests/golden/ where the reference is not available.
 *
 * Arithmetic contract (must match the reference build, Makefile:3-4):
 *   - IEEE-754 binary64, round-to-nearest, NO fused multiply-add
 *     (built -ffp-contract=off, no -march);
 *   - sums are accumulated strictly in the order written here;
 *   - int32 products/sums wrap modulo 2^32; `>>` on negatives is arithmetic;
 *   - the LU refinement residual is accumulated in x87 `long double`.
 */
#include "sla_oracle.h"

#include <float.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>

/* ---- format constants (reference src/include/private/SLAInternal.h:6-38,
 *      src/include/public/SLA.h:11-23) ------------------------------------ */
#define K_MAX_CHANNELS        8
#define K_HEADER_SIZE         43
#define K_BLOCK_HEADER_MIN    10
#define K_SYNC                0xFFFFu
#define K_LTM_MAX_PERIOD      256u
#define K_LTM_MIN_PITCH       3u
#define K_LTM_PERIOD_BITS     10u
#define K_MIN_BLOCK           2048u
#define K_SEARCH_DELTA        1024u
#define K_EMPH_SHIFT          5
#define K_RICE_PARAMS         2u
#define K_RICE_LOW_THRESHOLD  8u
#define K_QUOT_THRESHOLD      16u
#define K_PATH_PENALTY        300.0
#define K_EST_BLOCK_HEADER    50.0
#define K_RAW_THRESHOLD       0.95f
#define K_BIG_WEIGHT          ((double)(1UL << 24))
#define K_HDR_CRC_START       10
#define K_BLK_CRC_START       8
enum { BLK_COMPRESS = 0, BLK_SILENT = 1, BLK_RAW = 2 };

/* ======================================================================== */
/* integer helpers (reference src/include/private/SLAUtility.h:17-66)        */
/* ======================================================================== */

static inline uint32_t nlz32(uint32_t x) { return x ? (uint32_t)__builtin_clz(x) : 32u; }
static inline uint32_t log2ceil32(uint32_t x) { return 32u - nlz32(x - 1u); }
static inline uint32_t log2floor32(uint32_t x) { return 31u - nlz32(x); }
static inline uint32_t pow2ceil32(uint32_t x) { return 1u << log2ceil32(x); }
static inline int is_pow2(uint32_t x) { return (x & (x - 1u)) == 0; }
/* zig-zag fold: negative s -> -2s-1, else 2s, all modulo 2^32 (SLAUtility.h:37,39) */
static inline uint32_t fold32(int32_t s)
{
  uint32_t u = (uint32_t)s << 1;
  return (s < 0) ? ~u : u;
}
static inline int32_t unfold32(uint32_t u) { return (int32_t)(u >> 1) ^ -(int32_t)(u & 1u); }
static inline int32_t sign32(int32_t v) { return (v > 0) - (v < 0); }
static inline int32_t mul_wrap(int32_t a, int32_t b) { return (int32_t)((uint32_t)a * (uint32_t)b); }
static inline int32_t add_wrap(int32_t a, int32_t b) { return (int32_t)((uint32_t)a + (uint32_t)b); }
static inline int32_t sub_wrap(int32_t a, int32_t b) { return (int32_t)((uint32_t)a - (uint32_t)b); }
static inline int32_t shl_wrap(int32_t a, uint32_t s) { return (int32_t)((uint32_t)a << s); }

/* x86-64 cvttsd2si semantics for (int32_t)double, incl. the "integer
 * indefinite" result for NaN / out of range */
static inline int32_t f64_to_i32_x86(double v)
{
  if (!(v > -2147483649.0 && v < 2147483648.0)) { return INT32_MIN; }
  return (int32_t)v;
}

/* reference src/SLAUtility.c:436-439 */
static inline double round_half_away(double d)
{
  return (d >= 0.0) ? floor(d + 0.5) : -floor(-d + 0.5);
}
/* reference src/SLAUtility.c:442-447 */
static inline double log2_ref(double x) { return log(x) * 1.4426950408889634; }

/* ======================================================================== */
/* CRC16-IBM (reference src/SLAUtility.c:37-71, 322-339): reflected 0xA001,  */
/* initial value 0                                                           */
/* ======================================================================== */
static uint16_t g_crc_table[256];
static int g_crc_ready = 0;
static void crc_init(void)
{
  uint32_t i, k;
  for (i = 0; i < 256; i++) {
    uint32_t c = i;
    for (k = 0; k < 8; k++) { c = (c & 1u) ? ((c >> 1) ^ 0xA001u) : (c >> 1); }
    g_crc_table[i] = (uint16_t)c;
  }
  g_crc_ready = 1;
}
uint32_t slao_crc16(const uint8_t* data, uint32_t n)
{
  uint16_t crc = 0;
  uint32_t i;
  if (!g_crc_ready) { crc_init(); }
  for (i = 0; i < n; i++) { crc = (uint16_t)((crc >> 8) ^ g_crc_table[(crc ^ data[i]) & 0xFFu]); }
  return crc;
}

/* ======================================================================== */
/* MSB-first bit writer / reader                                             */
/* (reference src/include/private/SLABitStream.h:178-270, 333-350)           */
/* ======================================================================== */
typedef struct {
  uint8_t* base; size_t cap; size_t pos;
  uint64_t acc; uint32_t nacc;       /* nacc pending bits (0..7 between calls) in the low bits of acc */
  int overflow;
} bitw_t;

static void bw_open(bitw_t* w, uint8_t* mem, size_t cap) { w->base = mem; w->cap = cap; w->pos = 0; w->acc = 0; w->nacc = 0; w->overflow = 0; }
static inline void bw_put(bitw_t* w, uint32_t val, uint32_t nbits) /* 1 <= nbits <= 32 */
{
  uint64_t v = (nbits >= 32) ? (uint64_t)val : ((uint64_t)val & ((1ull << nbits) - 1ull));
  w->acc = (w->acc << nbits) | v;
  w->nacc += nbits;
  while (w->nacc >= 8) {
    w->nacc -= 8;
    if (w->pos < w->cap) { w->base[w->pos++] = (uint8_t)(w->acc >> w->nacc); } else { w->overflow = 1; }
  }
}
static inline void bw_zeros(bitw_t* w, uint32_t count)
{
  while (count >= 32) { bw_put(w, 0, 32); count -= 32; }
  if (count) { bw_put(w, 0, count); }
}
static inline void bw_align(bitw_t* w) { if (w->nacc) { bw_put(w, 0, 8 - w->nacc); } }
static inline size_t bw_tell(const bitw_t* w) { return w->pos; }

typedef struct { const uint8_t* base; size_t size; size_t pos; uint32_t cur; uint32_t left; } bitr_t;
static void br_open(bitr_t* r, const uint8_t* mem, size_t size) { r->base = mem; r->size = size; r->pos = 0; r->cur = 0; r->left = 0; }
static inline uint32_t br_bit(bitr_t* r)
{
  if (r->left == 0) { r->cur = (r->pos < r->size) ? r->base[r->pos] : 0u; r->pos++; r->left = 8; }
  r->left--;
  return (r->cur >> r->left) & 1u;
}
static inline uint32_t br_get(bitr_t* r, uint32_t nbits)
{
  uint32_t v = 0;
  while (nbits) {
    uint32_t take;
    if (r->left == 0) { r->cur = (r->pos < r->size) ? r->base[r->pos] : 0u; r->pos++; r->left = 8; }
    take = (nbits < r->left) ? nbits : r->left;
    r->left -= take;
    v = (v << take) | ((r->cur >> r->left) & ((1u << take) - 1u));
    nbits -= take;
  }
  return v;
}
/* zeros up to (and consuming) the next 1 bit; bounded so corrupt input cannot hang */
static inline uint32_t br_zero_run(bitr_t* r)
{
  uint32_t run = 0;
  while (br_bit(r) == 0) { run++; if (r->pos > r->size + 8) { break; } }
  return run;
}
static inline void br_align(bitr_t* r) { r->left = 0; }
static inline size_t br_tell(const bitr_t* r) { return r->pos; }

/* ======================================================================== */
/* windows (reference src/SLAUtility.c:99-189)                               */
/* ======================================================================== */
#define PI_REF 3.1415926535897932384626433832795029
int slao_window(uint32_t type, double* w, uint32_t n)
{
  uint32_t i;
  if (type > 4) { return -1; }
  if (type == 0) { for (i = 0; i < n; i++) { w[i] = 1.0; } return 0; }
  if (n == 1) { w[0] = 1.0; return 0; }
  for (i = 0; i < n; i++) {
    double x = (double)i / (n - 1);
    switch (type) {
      case 1: w[i] = sin(PI_REF * x); break;
      case 2: w[i] = 0.5f - 0.5f * cos(2.0f * PI_REF * x); break;
      case 3: w[i] = 0.42f - 0.5f * cos(2.0f * PI_REF * x) + 0.08f * cos(4.0f * PI_REF * x); break;
      default: w[i] = sin((PI_REF / 2.0f) * sin(PI_REF * x) * sin(PI_REF * x)); break;
    }
  }
  return 0;
}

/* ======================================================================== */
/* real FFT, Numerical-Recipes four1/realft evaluation order                 */
/* (reference src/SLAUtility.c:220-319); arrays are addressed 1-based as in  */
/* the published algorithm so that index arithmetic stays comparable         */
/* ======================================================================== */
static void nr_complex_fft(double* d, unsigned long nn, int isign)
{
  unsigned long n = nn << 1, i, j = 1, m, mmax, istep;
  for (i = 1; i < n; i += 2) {
    if (j > i) {
      double t;
      t = d[j]; d[j] = d[i]; d[i] = t;
      t = d[j + 1]; d[j + 1] = d[i + 1]; d[i + 1] = t;
    }
    m = n >> 1;
    while (m >= 2 && j > m) { j -= m; m >>= 1; }
    j += m;
  }
  for (mmax = 2; n > mmax; mmax = istep) {
    double theta = isign * (6.28318530717959 / (double)mmax);
    double wtemp = sin(0.5 * theta);
    double wpr = -2.0 * wtemp * wtemp;
    double wpi = sin(theta);
    double wr = 1.0, wi = 0.0;
    istep = mmax << 1;
    for (m = 1; m < mmax; m += 2) {
      for (i = m; i <= n; i += istep) {
        double tr, ti;
        j = i + mmax;
        tr = wr * d[j] - wi * d[j + 1];
        ti = wr * d[j + 1] + wi * d[j];
        d[j] = d[i] - tr;
        d[j + 1] = d[i + 1] - ti;
        d[i] += tr;
        d[i + 1] += ti;
      }
      wtemp = wr;
      wr = wtemp * wpr - wi * wpi + wr;
      wi = wi * wpr + wtemp * wpi + wi;
    }
  }
}

static void nr_real_fft(double* d, unsigned long n, int isign)
{
  unsigned long i, i1, i2, i3, i4, np3 = n + 3;
  double c1 = 0.5, c2, h1r, h1i, h2r, h2i, wr, wi, wpr, wpi, wtemp;
  double theta = 3.141592653589793 / (double)(n >> 1);
  if (isign == 1) { c2 = -0.5; nr_complex_fft(d, n >> 1, 1); }
  else { c2 = 0.5; theta = -theta; }
  wtemp = sin(0.5 * theta);
  wpr = -2.0 * wtemp * wtemp;
  wpi = sin(theta);
  wr = 1.0 + wpr;
  wi = wpi;
  for (i = 2; i <= (n >> 2); i++) {
    i1 = i + i - 1; i2 = 1 + i1; i3 = np3 - i2; i4 = 1 + i3;
    h1r = c1 * (d[i1] + d[i3]);
    h1i = c1 * (d[i2] - d[i4]);
    h2r = -c2 * (d[i2] + d[i4]);
    h2i = c2 * (d[i1] - d[i3]);
    d[i1] = h1r + wr * h2r - wi * h2i;
    d[i2] = h1i + wr * h2i + wi * h2r;
    d[i3] = h1r - wr * h2r + wi * h2i;
    d[i4] = -h1i + wr * h2i + wi * h2r;
    wtemp = wr;
    wr = wtemp * wpr - wi * wpi + wr;
    wi = wi * wpr + wtemp * wpi + wi;
  }
  if (isign == 1) {
    h1r = d[1];
    d[1] = h1r + d[2];
    d[2] = h1r - d[2];
  } else {
    h1r = d[1];
    d[1] = c1 * (h1r + d[2]);
    d[2] = c1 * (h1r - d[2]);
    nr_complex_fft(d, n >> 1, -1);
  }
}

void slao_fft(double* data, uint32_t n, int32_t sign) { nr_real_fft(data - 1, n, (int)sign); }

/* ======================================================================== */
/* LU solve with scaled partial pivoting and iterative refinement            */
/* (reference src/SLAUtility.c:487-674); dim <= 8 here                       */
/* ======================================================================== */
#define LU_MAX 8
static int lu_factor(double A[LU_MAX][LU_MAX], uint32_t dim, uint32_t* perm, double* scale)
{
  uint32_t row, col, k, imax;
  double big, sum;
  for (row = 0; row < dim; row++) {
    big = 0.0;
    for (col = 0; col < dim; col++) { if (fabs(A[row][col]) > big) { big = fabs(A[row][col]); } }
    if (fabs(big) <= FLT_EPSILON) { return -1; }
    scale[row] = 1.0f / big;
  }
  for (col = 0; col < dim; col++) {
    for (row = 0; row < col; row++) {
      sum = A[row][col];
      for (k = 0; k < row; k++) { sum -= A[row][k] * A[k][col]; }
      A[row][col] = sum;
    }
    big = 0.0;
    imax = row;
    for (row = col; row < dim; row++) {
      sum = A[row][col];
      for (k = 0; k < col; k++) { sum -= A[row][k] * A[k][col]; }
      A[row][col] = sum;
      if ((scale[row] * fabs(sum)) >= big) { big = scale[row] * fabs(sum); imax = row; }
    }
    if (col != imax) {
      for (k = 0; k < dim; k++) { double t = A[imax][k]; A[imax][k] = A[col][k]; A[col][k] = t; }
      scale[imax] = scale[col];
    }
    perm[col] = imax;
    if (fabs(A[col][col]) <= FLT_EPSILON) { return -1; }
    if (col != dim - 1) {
      double inv = 1.0f / A[col][col];
      for (row = col + 1; row < dim; row++) { A[row][col] *= inv; }
    }
  }
  return 0;
}

static void lu_substitute(double A[LU_MAX][LU_MAX], double* b, uint32_t dim, const uint32_t* perm)
{
  uint32_t row, col, first_nz = 0;
  double sum;
  for (row = 0; row < dim; row++) {
    uint32_t pv = perm[row];
    sum = b[pv];
    b[pv] = b[row];
    if (first_nz != 0) {
      for (col = first_nz; col < row; col++) { sum -= A[row][col] * b[col]; }
    } else if (sum != 0.0) {
      first_nz = row;
    }
    b[row] = sum;
  }
  for (row = dim; row-- > 0;) {
    sum = b[row];
    for (col = row + 1; col < dim; col++) { sum -= A[row][col] * b[col]; }
    b[row] = sum / A[row][row];
  }
}

/* A row-major dim x dim, b in/out */
int slao_lesolve(const double* Ain, double* b, uint32_t dim, uint32_t iters)
{
  double A[LU_MAX][LU_MAX], x[LU_MAX], err[LU_MAX], scale[LU_MAX];
  uint32_t perm[LU_MAX], row, col, it;
  if (dim == 0 || dim > LU_MAX) { return -1; }
  for (row = 0; row < dim; row++) { for (col = 0; col < dim; col++) { A[row][col] = Ain[row * dim + col]; } }
  memcpy(x, b, sizeof(double) * dim);
  if (lu_factor(A, dim, perm, scale) != 0) { return -1; }
  lu_substitute(A, x, dim, perm);
  for (it = 0; it < iters; it++) {
    for (row = 0; row < dim; row++) {
      long double e = -b[row];      /* x87 extended accumulation, as the reference */
      for (col = 0; col < dim; col++) { e += Ain[row * dim + col] * x[col]; }
      err[row] = (double)e;
    }
    lu_substitute(A, err, dim, perm);
    for (row = 0; row < dim; row++) { x[row] -= err[row]; }
  }
  memcpy(b, x, sizeof(double) * dim);
  return 0;
}

/* ======================================================================== */
/* A1: sample autocorrelation with the reference's paired-product order      */
/* (reference src/SLAPredictor.c:331-388).  r has nlags entries.             */
/* ======================================================================== */
int slao_autocorr(const double* x, uint32_t n, double* r, uint32_t nlags)
{
  uint32_t lag, i, l;
  if (x == NULL || r == NULL) { return 2; }
  if (nlags > n) { nlags = n; }
  for (i = 0; i < nlags; i++) { r[i] = 0.0; }
  if (nlags == 0) { return 0; }
  for (i = 0; i < n; i++) { r[0] += x[i] * x[i]; }
  for (lag = 1; lag < nlags; lag++) {
    const uint32_t lag2 = lag << 1;
    uint32_t groups = ((3 * lag) < n) ? (1 + (n - 3 * lag) / lag2) : 0;
    uint32_t span = groups * lag2;
    double acc = r[lag];
    /* x[l+lag+i] multiplies both neighbours at distance lag: one product per pair */
    for (i = 0; i < lag; i++) {
      for (l = 0; l < span; l += lag2) {
        acc += x[l + lag + i] * (x[l + i] + x[l + lag2 + i]);
      }
    }
    for (i = 0; i < (n - span - lag); i++) {
      acc += x[span + lag + i] * x[span + i];
    }
    r[lag] = acc;
  }
  return 0;
}

/* ======================================================================== */
/* A2: Levinson-Durbin -> LPC a[] and PARCOR k[]                             */
/* (reference src/SLAPredictor.c:253-328)                                    */
/* ======================================================================== */
#define LPC_MAX_ORDER 256
int slao_levinson(const double* r, uint32_t order, double* lpc, double* parcor)
{
  double a[LPC_MAX_ORDER + 2], e[LPC_MAX_ORDER + 2], u[LPC_MAX_ORDER + 2], v[LPC_MAX_ORDER + 2];
  uint32_t d, i;
  if (order > LPC_MAX_ORDER || r == NULL || lpc == NULL || parcor == NULL) { return 2; }
  if (fabs(r[0]) < FLT_EPSILON) {
    for (i = 0; i < order + 1; i++) { lpc[i] = parcor[i] = 0.0; }
    return 0;
  }
  for (i = 0; i < order + 2; i++) { a[i] = u[i] = v[i] = 0.0; }
  a[0] = 1.0;
  e[0] = r[0];
  a[1] = -r[1] / r[0];
  parcor[0] = 0.0;
  parcor[1] = r[1] / e[0];
  e[1] = r[0] + r[1] * a[1];
  u[0] = 1.0; u[1] = 0.0;
  v[0] = 0.0; v[1] = 1.0;
  for (d = 1; d < order; d++) {
    double gamma = 0.0;
    for (i = 0; i < d + 1; i++) { gamma += a[i] * r[d + 1 - i]; }
    gamma /= (-e[d]);
    e[d + 1] = (1.0 - gamma * gamma) * e[d];
    for (i = 0; i < d; i++) { u[i + 1] = v[d - i] = a[i + 1]; }
    u[0] = 1.0; u[d + 1] = 0.0;
    v[0] = 0.0; v[d + 1] = 1.0;
    for (i = 0; i < d + 2; i++) { a[i] = u[i] + gamma * v[i]; }
    parcor[d + 1] = -gamma;
  }
  memcpy(lpc, a, sizeof(double) * (order + 1));
  return 0;
}

/* reference src/SLAPredictor.c:189-250.  `r` is the calculator's autocorrelation scratch: the
 * reference keeps it inside the handle across calls and, when n == order exactly, the lag-`order`
 * entry is not rewritten (src/SLAPredictor.c:344-346) and the recursion consumes the value left by
 * the previous call.  Callers that own a handle pass its scratch; one-shot callers get zeros. */
static int parcor_with_scratch(double* r, const double* x, uint32_t n, uint32_t order, double* parcor)
{
  double lpc[LPC_MAX_ORDER + 1], k[LPC_MAX_ORDER + 2];
  uint32_t i;
  if (x == NULL || parcor == NULL) { return 2; }
  if (order > LPC_MAX_ORDER) { return 3; }
  slao_autocorr(x, n, r, order + 1);
  if (n < order) {
    for (i = 0; i < order + 1; i++) { parcor[i] = 0.0; }
    return 0;
  }
  if (slao_levinson(r, order, lpc, k) != 0) { return 4; }
  memcpy(parcor, k, sizeof(double) * (order + 1));
  return 0;
}

int slao_parcor(const double* x, uint32_t n, uint32_t order, double* parcor)
{
  double r[LPC_MAX_ORDER + 1];
  memset(r, 0, sizeof(r));
  return parcor_with_scratch(r, x, n, order, parcor);
}

/* ======================================================================== */
/* A3: entropy estimate in bytes/sample (reference src/SLAPredictor.c:416-468) */
/* ======================================================================== */
static double code_length_from_power(double sumsq, uint32_t n, uint32_t bps, const double* parcor, uint32_t order)
{
  double p = sumsq, lvar = 0.0, len;
  uint32_t ord;
  p *= ldexp(1.0, (int)(2 * (bps - 1)));
  if (fabs(p) <= FLT_MIN) { return 0.0; }
  p = log2_ref(p) - log2_ref((double)n);
  for (ord = 1; ord <= order; ord++) { lvar += log2_ref(1.0 - parcor[ord] * parcor[ord]); }
  len = 1.9426950408889634 + 0.5f * (p + lvar);
  len /= 8;
  if (len <= 0) { return 1.0f / 8; }
  return len;
}

int slao_code_length(const double* x, uint32_t n, uint32_t bps, const double* parcor, uint32_t order, double* out)
{
  double s = 0.0;
  uint32_t i;
  if (x == NULL || parcor == NULL || out == NULL) { return 2; }
  for (i = 0; i < n; i++) { s += x[i] * x[i]; }
  *out = code_length_from_power(s, n, bps, parcor, order);
  return 0;
}

/* ======================================================================== */
/* A5: PARCOR analysis lattice, zero initial state                           */
/* (reference src/SLAPredictor.c:557-607) and its inverse (:610-740)         */
/* ======================================================================== */
static inline int32_t lattice_term(int32_t k, int32_t v) { return add_wrap(mul_wrap(k, v), 1 << 14) >> 15; }

int slao_lattice_predict(const int32_t* x, uint32_t n, const int32_t* kint, uint32_t order, int32_t* res)
{
  int32_t fwd[LPC_MAX_ORDER + 1], bwd[LPC_MAX_ORDER + 1];
  uint32_t s, m;
  if (x == NULL || kint == NULL || res == NULL) { return 2; }
  if (order > LPC_MAX_ORDER) { return 3; }
  memset(bwd, 0, sizeof(bwd));
  for (s = 0; s < n; s++) {
    fwd[0] = x[s];
    for (m = 1; m <= order; m++) { fwd[m] = sub_wrap(fwd[m - 1], lattice_term(kint[m], bwd[m - 1])); }
    for (m = order; m >= 1; m--) { bwd[m] = sub_wrap(bwd[m - 1], lattice_term(kint[m], fwd[m - 1])); }
    bwd[0] = x[s];
    res[s] = fwd[order];
  }
  return 0;
}

int slao_lattice_synth(const int32_t* res, uint32_t n, const int32_t* kint, uint32_t order, int32_t* out)
{
  int32_t bwd[LPC_MAX_ORDER + 1];
  uint32_t s, m;
  if (res == NULL || kint == NULL || out == NULL) { return 2; }
  if (order > LPC_MAX_ORDER) { return 3; }
  memset(bwd, 0, sizeof(bwd));
  for (s = 0; s < n; s++) {
    int32_t f = res[s];
    for (m = order; m >= 1; m--) {
      f = add_wrap(f, lattice_term(kint[m], bwd[m - 1]));
      bwd[m] = sub_wrap(bwd[m - 1], lattice_term(kint[m], f));
    }
    out[s] = f;
    bwd[0] = f;
  }
  return 0;
}

/* ======================================================================== */
/* A4: pre-/de-emphasis (reference src/SLAPredictor.c:1741-1813)             */
/* ======================================================================== */
int slao_preemph_i32(int32_t* data, uint32_t n)
{
  const int32_t numer = (1 << K_EMPH_SHIFT) - 1;
  int32_t prev = 0;
  uint32_t i;
  if (data == NULL) { return 2; }
  for (i = 0; i < n; i++) {
    int32_t cur = data[i];
    data[i] = sub_wrap(cur, mul_wrap(prev, numer) >> K_EMPH_SHIFT);
    prev = cur;
  }
  return 0;
}

int slao_deemph_i32(int32_t* data, uint32_t n)
{
  const int32_t numer = (1 << K_EMPH_SHIFT) - 1;
  uint32_t i;
  if (data == NULL) { return 2; }
  for (i = 1; i < n; i++) { data[i] = add_wrap(data[i], mul_wrap(data[i - 1], numer) >> K_EMPH_SHIFT); }
  return 0;
}

void slao_preemph_f64(double* data, uint32_t n)
{
  const double coef = (ldexp(1.0, K_EMPH_SHIFT) - 1.0) * ldexp(1.0, -K_EMPH_SHIFT);
  double prev = 0.0;
  uint32_t i;
  for (i = 0; i < n; i++) {
    double cur = data[i];
    data[i] -= prev * coef;
    prev = cur;
  }
}

/* A6 helper: reference src/SLAUtility.c:677-696 */
uint32_t slao_bitwidth(const int32_t* data, uint32_t n)
{
  uint32_t maxabs = 0, i;
  for (i = 0; i < n; i++) {
    uint32_t a = (data[i] > 0) ? (uint32_t)data[i] : (0u - (uint32_t)data[i]);
    if (a > maxabs) { maxabs = a; }
  }
  return (maxabs > 0) ? (log2ceil32(maxabs) + 1u) : 1u;
}

/* ======================================================================== */
/* long-term (pitch) predictor: FFT analysis (reference                      */
/* src/SLAPredictor.c:791-980) and the integer tap filter (:1031-1108)       */
/* ======================================================================== */
/* returns SLAPredictorApiResult numbering: 0 OK, 2 invalid arg, 3 exceed order, 4 failed to calculate */
int slao_ltm_analyze(const int32_t* res, uint32_t n, uint32_t fft_size, uint32_t max_taps, uint32_t ntaps,
                     uint32_t* pitch, double* coef, double* autocorr_out)
{
  double* ac;
  uint32_t i, num_peak = 0, cand[K_LTM_MAX_PERIOD], tmp_pitch;
  double max_peak = 0.0;
  int ret = 0;
  if (res == NULL || pitch == NULL || coef == NULL) { return 2; }
  if (!(ntaps & 1u)) { return 2; }
  if (ntaps > max_taps || ntaps > LU_MAX) { return 3; }
  if (fft_size & (fft_size - 1)) { return 2; }
  if (2 * n > fft_size) { return 2; }
  ac = (double*)malloc(sizeof(double) * fft_size);
  for (i = 0; i < fft_size; i++) { ac[i] = (i < n) ? (double)res[i] * ldexp(1.0, -31) : 0.0; }
  slao_fft(ac, fft_size, 1);
  ac[0] *= ac[0];
  ac[1] *= ac[1];
  for (i = 1; i < fft_size / 2; i++) {
    double re = ac[2 * i], im = ac[2 * i + 1];
    ac[2 * i] = re * re + im * im;
    ac[2 * i + 1] = 0.0;
  }
  slao_fft(ac, fft_size, -1);
  if (autocorr_out != NULL) { memcpy(autocorr_out, ac, sizeof(double) * fft_size); }

  if (fabs(ac[0]) <= FLT_MIN) {
    *pitch = 0;
    for (i = 0; i < ntaps; i++) { coef[i] = 0.0; }
    goto done;
  }
  /* peak picking between upward and downward zero crossings */
  i = 1;
  while (i < K_LTM_MAX_PERIOD && num_peak < K_LTM_MAX_PERIOD) {
    uint32_t start, end, j, best = 0;
    double best_val = 0.0;
    for (start = i; start < K_LTM_MAX_PERIOD; start++) {
      if (ac[start - 1] < 0.0 && ac[start] > 0.0) { break; }
    }
    for (end = start + 1; end < K_LTM_MAX_PERIOD; end++) {
      if (ac[end] > 0.0 && ac[end + 1] < 0.0) { break; }
    }
    for (j = start; j <= end; j++) {
      if (ac[j] > ac[j - 1] && ac[j] > ac[j + 1] && ac[j] > best_val) { best = j; best_val = ac[j]; }
    }
    if (best != 0) {
      cand[num_peak++] = best;
      if (best_val > max_peak) { max_peak = best_val; }
    }
    i = end + 1;
  }
  if (num_peak == 0) { ret = 4; goto done; }
  for (i = 0; i < num_peak; i++) { if (ac[cand[i]] >= 1.0f * max_peak) { break; } }
  tmp_pitch = cand[i];
  if (tmp_pitch < (ntaps / 2 + 1)) { ret = 4; goto done; }
  {
    double R[LU_MAX * LU_MAX], vec[LU_MAX], asum = 0.0;
    uint32_t j, k;
    for (j = 0; j < ntaps; j++) {
      for (k = 0; k < ntaps; k++) { R[j * ntaps + k] = ac[(j >= k) ? (j - k) : (k - j)]; }
    }
    for (j = 0; j < ntaps; j++) { vec[j] = ac[j + tmp_pitch - ntaps / 2]; }
    if (slao_lesolve(R, vec, ntaps, 2) != 0) { ret = 4; goto done; }
    for (j = 0; j < ntaps; j++) { asum += fabs(vec[j]); }
    if (asum >= 1.0) {
      for (j = 0; j < ntaps; j++) { vec[j] = 0.0; }
      vec[ntaps / 2] = ac[tmp_pitch] / ac[0];
    }
    *pitch = tmp_pitch;
    for (j = 0; j < ntaps; j++) { coef[j] = vec[j]; }
  }
done:
  free(ac);
  return ret;
}

/* Pitch-delayed FIR in Q31 with int64 accumulation; the first pitch+ntaps/2
 * samples pass through (reference ring buffer, src/SLAPredictor.c:1063-1099) */
static int ltm_run(const int32_t* in, uint32_t n, uint32_t pitch, const int32_t* coef, uint32_t ntaps,
                   int32_t* out, int predict)
{
  uint32_t s, j, delay;
  if (in == NULL || coef == NULL || out == NULL) { return 2; }
  memcpy(out, in, sizeof(int32_t) * n);
  if (pitch == 0) { return 0; }
  delay = pitch + (ntaps >> 1);
  for (s = delay; s < n; s++) {
    const int32_t* hist = predict ? in : out;
    int64_t acc = (int64_t)1 << 30;
    for (j = 0; j < ntaps; j++) { acc += (int64_t)coef[j] * hist[s - delay + j]; }
    acc >>= 31;
    out[s] = predict ? sub_wrap(out[s], (int32_t)acc) : add_wrap(out[s], (int32_t)acc);
  }
  return 0;
}
int slao_ltm_predict(const int32_t* in, uint32_t n, uint32_t pitch, const int32_t* coef, uint32_t ntaps, int32_t* out)
{ return ltm_run(in, n, pitch, coef, ntaps, out, 1); }
int slao_ltm_synth(const int32_t* in, uint32_t n, uint32_t pitch, const int32_t* coef, uint32_t ntaps, int32_t* out)
{ return ltm_run(in, n, pitch, coef, ntaps, out, 0); }

/* ======================================================================== */
/* sign-log LMS cascade (reference src/SLAPredictor.c:121-145, 1202-1463):   */
/* two adaptive FIRs, one on past inputs and one on past *predictions*,      */
/* step = sign(e)*sign(history)*(ceil(log2(|e|+1))>>1)                       */
/* ======================================================================== */
#define LMS_MAX 64
static int lms_run(const int32_t* in, uint32_t n, uint32_t order, int32_t* out, int predict)
{
  int32_t cf[LMS_MAX], ci[LMS_MAX], hx[LMS_MAX], hp[LMS_MAX];  /* coefs, input history, prediction history */
  int32_t sx[LMS_MAX], sp[LMS_MAX];                             /* signs of the histories */
  uint32_t s, i, head, mask, warm;
  if (in == NULL || out == NULL) { return 2; }
  if (order > LMS_MAX || order < 4 || !is_pow2(order)) { return 3; }
  mask = order - 1;
  memset(cf, 0, sizeof(cf)); memset(ci, 0, sizeof(ci));
  memset(hx, 0, sizeof(hx)); memset(hp, 0, sizeof(hp));
  memset(sx, 0, sizeof(sx)); memset(sp, 0, sizeof(sp));
  memcpy(out, in, sizeof(int32_t) * n);
  warm = (n < order) ? n : order;
  if (n < order) { return 0; }
  /* warm-up: both histories are primed with the first `order` inputs, newest at slot 0 */
  for (i = 0; i < warm; i++) {
    hx[i] = hp[i] = in[warm - 1 - i];
    sx[i] = sp[i] = sign32(in[warm - 1 - i]);
  }
  head = 0;   /* slot of the newest entry; entry i-th most recent is (head + i) & mask */
  for (s = warm; s < n; s++) {
    int32_t pred = 1 << 9, err, step;
    uint32_t lg;
    for (i = 0; i < order; i++) {
      uint32_t slot = (head + i) & mask;
      pred = add_wrap(pred, mul_wrap(cf[i], hx[slot]));
      pred = add_wrap(pred, mul_wrap(ci[i], hp[slot]));
    }
    pred >>= 10;
    if (predict) { out[s] = sub_wrap(out[s], pred); err = out[s]; }
    else { err = out[s]; out[s] = add_wrap(out[s], pred); }
    lg = 32u - nlz32((err > 0) ? (uint32_t)err : (0u - (uint32_t)err));  /* ceil(log2(|e|+1)) */
    step = sign32(err) * (int32_t)((lg << 4) >> 5);
    for (i = 0; i < order; i++) {
      uint32_t slot = (head + i) & mask;
      cf[i] += step * sx[slot];
      ci[i] += step * sp[slot];
    }
    head = (head - 1) & mask;
    hx[head] = predict ? in[s] : out[s];
    hp[head] = pred;
    sx[head] = sign32(hx[head]);
    sp[head] = sign32(pred);
  }
  return 0;
}
int slao_lms_predict(const int32_t* in, uint32_t n, uint32_t order, int32_t* out) { return lms_run(in, n, order, out, 1); }
int slao_lms_synth(const int32_t* in, uint32_t n, uint32_t order, int32_t* out) { return lms_run(in, n, order, out, 0); }

/* ======================================================================== */
/* A8: block-partition search (reference src/SLAPredictor.c:1521-1705)       */
/* ======================================================================== */
#define MAX_NODES 80
int slao_dijkstra(const double* adj, uint32_t nodes, uint32_t start, uint32_t goal, double* min_cost, uint32_t* path)
{
  double cost[MAX_NODES];
  uint8_t used[MAX_NODES];
  uint32_t i, target = 0, guard;
  if (nodes > MAX_NODES || adj == NULL || min_cost == NULL || path == NULL) { return 2; }
  for (i = 0; i < nodes; i++) { used[i] = 0; path[i] = 0xFFFFFFFFu; cost[i] = K_BIG_WEIGHT; }
  cost[start] = 0.0;
  for (guard = 0; guard <= nodes + 1; guard++) {
    double best = K_BIG_WEIGHT;
    for (i = 0; i < nodes; i++) { if (!used[i] && cost[i] < best) { best = cost[i]; target = i; } }
    if (target == goal) { *min_cost = cost[goal]; return 0; }
    for (i = 0; i < nodes; i++) {
      if (cost[i] > adj[target * nodes + i] + cost[target]) {
        cost[i] = adj[target * nodes + i] + cost[target];
        path[i] = target;
      }
    }
    used[target] = 1;
  }
  return 4;
}

static uint32_t num_nodes_for(uint32_t n, uint32_t delta) { return (n + delta - 1) / delta + 1; }

static int partition_search_scratch(double* rscr, const double* data, uint32_t nch, uint32_t n, uint32_t min_blk,
                                    uint32_t delta, uint32_t max_blk, uint32_t bps, uint32_t order,
                                    uint32_t* num_parts, uint32_t* parts)
{
  double adj[MAX_NODES * MAX_NODES], parcor[LPC_MAX_ORDER + 1], total;
  uint32_t path[MAX_NODES], nodes, i, j, ch, count, node;
  if (data == NULL || num_parts == NULL || parts == NULL) { return 2; }
  nodes = num_nodes_for(n, delta);
  if (nodes > MAX_NODES) { return 3; }
  for (i = 0; i < nodes; i++) {
    for (j = 0; j < nodes; j++) {
      uint32_t len, off;
      double est = 0.0;
      adj[i * nodes + j] = K_BIG_WEIGHT;
      if (j <= i) { continue; }
      off = i * delta;
      len = (j - i) * delta;
      if (len > n - off) { len = n - off; }
      if (len < min_blk || len > max_blk) { continue; }
      for (ch = 0; ch < nch; ch++) {
        const double* x = &data[(size_t)ch * n + off];
        double per_sample;
        if (parcor_with_scratch(rscr, x, len, order, parcor) != 0) { return 4; }
        slao_code_length(x, len, bps, parcor, order, &per_sample);
        est += len * per_sample;
      }
      est += K_EST_BLOCK_HEADER;
      est += K_PATH_PENALTY;
      adj[i * nodes + j] = est;
    }
  }
  if (slao_dijkstra(adj, nodes, 0, nodes - 1, &total, path) != 0) { return 4; }
  count = 0;
  for (node = nodes - 1; node != 0; node = path[node]) {
    if (path[node] >= node) { return 4; }
    count++;
  }
  node = nodes - 1;
  for (i = 0; i < count; i++) {
    uint32_t off = path[node] * delta, len = (node - path[node]) * delta;
    if (len > n - off) { len = n - off; }
    parts[count - i - 1] = len;
    node = path[node];
  }
  *num_parts = count;
  return 0;
}

int slao_partition_search(const double* data, uint32_t nch, uint32_t n, uint32_t min_blk, uint32_t delta,
                          uint32_t max_blk, uint32_t bps, uint32_t order, uint32_t* num_parts, uint32_t* parts)
{
  double r[LPC_MAX_ORDER + 1];
  memset(r, 0, sizeof(r));
  return partition_search_scratch(r, data, nch, n, min_blk, delta, max_blk, bps, order, num_parts, parts);
}

/* ======================================================================== */
/* entropy coder: recursive Rice with 2 adaptive parameters, Golomb for      */
/* small residuals, gamma escape (reference src/SLACoder.c:9-31, 45-318,     */
/* 361-506)                                                                  */
/* ======================================================================== */
typedef uint64_t ricep_t;   /* 8 fractional bits */
static inline uint32_t rp_int(ricep_t f) { return (uint32_t)((f + 128u) >> 8); }
static inline uint32_t rp_get(ricep_t f) { uint32_t v = rp_int(f); return v > 1u ? v : 1u; }
static inline uint32_t rp_rice(ricep_t f) { uint32_t v = rp_int(f >> 1); return pow2ceil32(v > 1u ? v : 1u); }
static inline ricep_t rp_set(uint32_t v) { return (ricep_t)(uint32_t)(v << 8); }
static inline void rp_update(ricep_t* f, uint32_t code)
{
  /* 119/128 * old + 9/128 * code, the code term evaluated in 32-bit unsigned arithmetic */
  *f = (119u * (*f) + (uint64_t)(uint32_t)(9u * (uint32_t)(code << 8)) + 64u) >> 7;
}

static void put_unary(bitw_t* w, uint32_t q) { bw_zeros(w, q); bw_put(w, 1, 1); }

static void golomb_put(bitw_t* w, uint32_t m, uint32_t val)
{
  uint32_t quot = val / m, rest = val % m;
  put_unary(w, quot);
  if (is_pow2(m)) {
    if (m > 1) { bw_put(w, rest, log2ceil32(m)); }
  } else {
    uint32_t b = log2ceil32(m), cut = (1u << b) - m;
    if (rest < cut) { bw_put(w, rest, b - 1); } else { bw_put(w, rest + cut, b); }
  }
}
static uint32_t golomb_get(bitr_t* r, uint32_t m)
{
  uint32_t quot = br_zero_run(r), rest, b, cut;
  if (is_pow2(m)) {
    rest = (m > 1) ? br_get(r, log2ceil32(m)) : 0;
    return quot * m + rest;
  }
  b = log2ceil32(m); cut = (1u << b) - m;
  rest = br_get(r, b - 1);
  if (rest < cut) { return quot * m + rest; }
  rest = (rest << 1) + br_get(r, 1);
  return quot * m + rest - cut;
}
static void gamma_put(bitw_t* w, uint32_t val)
{
  uint32_t nd;
  if (val == 0) { bw_put(w, 1, 1); return; }
  nd = log2ceil32(val + 2);
  bw_put(w, 0, nd - 1);
  bw_put(w, val + 1, nd);
}
static uint32_t gamma_get(bitr_t* r)
{
  uint32_t nd = br_zero_run(r) + 1;
  if (nd == 1) { return 0; }
  return (1u << (nd - 1)) + br_get(r, nd - 1) - 1u;
}
static inline void rest_put(bitw_t* w, uint32_t val, uint32_t m) { if (m != 1) { bw_put(w, val & (m - 1), log2ceil32(m)); } }
static inline uint32_t rest_get(bitr_t* r, uint32_t m) { return (m == 1) ? 0 : br_get(r, log2ceil32(m)); }

static void rrice_put(bitw_t* w, ricep_t* prm, uint32_t val)
{
  uint32_t i, v = val;
  for (i = 0; i < K_RICE_PARAMS - 1; i++) {
    uint32_t m = rp_rice(prm[i]);
    if (v < m) {
      put_unary(w, i);
      rest_put(w, v, m);
      rp_update(&prm[i], v);
      return;
    }
    rp_update(&prm[i], v);
    v -= m;
  }
  {
    uint32_t m = rp_rice(prm[i]), q = i + v / m;
    if (q < K_QUOT_THRESHOLD) { put_unary(w, q); }
    else { put_unary(w, K_QUOT_THRESHOLD); gamma_put(w, q - K_QUOT_THRESHOLD); }
    rest_put(w, v, m);
    rp_update(&prm[i], v);
  }
}
static uint32_t rrice_get(bitr_t* r, ricep_t* prm)
{
  uint32_t quot = br_zero_run(r), val = 0, i, tmp;
  for (i = 0; i < quot && i < K_RICE_PARAMS - 1; i++) { val += rp_rice(prm[i]); }
  if (quot < K_RICE_PARAMS - 1) {
    val += rest_get(r, rp_rice(prm[i]));
  } else {
    uint32_t m = rp_rice(prm[i]);
    if (quot == K_QUOT_THRESHOLD) { quot += gamma_get(r); }
    val += m * (quot - (K_RICE_PARAMS - 1)) + rest_get(r, m);
  }
  tmp = val;
  for (i = 0; i <= quot && i < K_RICE_PARAMS; i++) {
    uint32_t m = rp_rice(prm[i]);
    rp_update(&prm[i], tmp);
    tmp -= m;
  }
  return val;
}

/* A7: mean of folded residual, at least 1 (reference src/SLACoder.c:361-385) */
static uint32_t rice_init_1ch(const int32_t* res, uint32_t n)
{
  uint64_t sum = 0, mean;
  uint32_t i;
  for (i = 0; i < n; i++) { sum += fold32(res[i]); }
  mean = sum / n;
  return (uint32_t)(mean > 1 ? mean : 1);
}
void slao_rice_init(const int32_t* res, uint32_t nch, uint32_t n, uint32_t* rice_init)
{
  uint32_t ch;
  for (ch = 0; ch < nch; ch++) { rice_init[ch] = rp_get(rp_set(rice_init_1ch(&res[(size_t)ch * n], n))); }
}

/* channel-interleaved residual body (reference src/SLACoder.c:429-467) */
static void put_residuals(bitw_t* w, const int32_t* const* res, uint32_t nch, uint32_t n, const uint32_t* init)
{
  ricep_t prm[K_MAX_CHANNELS][K_RICE_PARAMS], first[K_MAX_CHANNELS];
  uint64_t avg = 0;
  uint32_t ch, s, i;
  for (ch = 0; ch < nch; ch++) {
    first[ch] = rp_set(init[ch]);
    for (i = 0; i < K_RICE_PARAMS; i++) { prm[ch][i] = first[ch]; }
    avg += rp_get(first[ch]);
  }
  avg /= nch;
  if (avg > K_RICE_LOW_THRESHOLD) {
    for (s = 0; s < n; s++) { for (ch = 0; ch < nch; ch++) { rrice_put(w, prm[ch], fold32(res[ch][s])); } }
  } else {
    for (s = 0; s < n; s++) { for (ch = 0; ch < nch; ch++) { golomb_put(w, rp_get(first[ch]), fold32(res[ch][s])); } }
  }
}
static void get_residuals(bitr_t* r, int32_t* const* res, uint32_t nch, uint32_t n, const uint32_t* init)
{
  ricep_t prm[K_MAX_CHANNELS][K_RICE_PARAMS], first[K_MAX_CHANNELS];
  uint64_t avg = 0;
  uint32_t ch, s, i;
  for (ch = 0; ch < nch; ch++) {
    first[ch] = rp_set(init[ch]);
    for (i = 0; i < K_RICE_PARAMS; i++) { prm[ch][i] = first[ch]; }
    avg += rp_get(first[ch]);
  }
  avg /= nch;
  if (avg > K_RICE_LOW_THRESHOLD) {
    for (s = 0; s < n; s++) { for (ch = 0; ch < nch; ch++) { res[ch][s] = unfold32(rrice_get(r, prm[ch])); } }
  } else {
    for (s = 0; s < n; s++) { for (ch = 0; ch < nch; ch++) { res[ch][s] = unfold32(golomb_get(r, rp_get(first[ch]))); } }
  }
}

uint32_t slao_code_residual(const int32_t* res, uint32_t nch, uint32_t n, uint32_t bps, uint8_t* out, uint32_t cap)
{
  const int32_t* ptr[K_MAX_CHANNELS];
  uint32_t init[K_MAX_CHANNELS], ch;
  bitw_t w;
  for (ch = 0; ch < nch; ch++) { ptr[ch] = &res[(size_t)ch * n]; init[ch] = rice_init_1ch(ptr[ch], n); }
  bw_open(&w, out, cap);
  for (ch = 0; ch < nch; ch++) { bw_put(&w, rp_get(rp_set(init[ch])), bps); }
  bw_align(&w);
  put_residuals(&w, ptr, nch, n, init);
  bw_align(&w);
  return (uint32_t)bw_tell(&w);
}
void slao_decode_residual(const uint8_t* in, uint32_t size, uint32_t nch, uint32_t n, uint32_t bps, int32_t* res)
{
  int32_t* ptr[K_MAX_CHANNELS];
  uint32_t init[K_MAX_CHANNELS], ch;
  bitr_t r;
  br_open(&r, in, size);
  for (ch = 0; ch < nch; ch++) { ptr[ch] = &res[(size_t)ch * n]; init[ch] = br_get(&r, bps); }
  br_align(&r);
  get_residuals(&r, ptr, nch, n, init);
}

/* ======================================================================== */
/* encoder (reference src/SLAEncoder.c)                                      */
/* ======================================================================== */
typedef struct {
  sla_flat_params p;
  uint32_t lshift;
  uint32_t fft_size;
  double*  xd[K_MAX_CHANNELS];      /* analysis signal            */
  int32_t* xi[K_MAX_CHANNELS];      /* integer signal (shifted, MS) */
  int32_t* res[K_MAX_CHANNELS];
  int32_t* tmp[K_MAX_CHANNELS];
  double*  window; uint32_t window_n; uint32_t window_type;
  /* last block's per-channel parameters (for tracing) */
  double   parcor[K_MAX_CHANNELS][LPC_MAX_ORDER + 1];
  int32_t  code[K_MAX_CHANNELS][LPC_MAX_ORDER + 1];
  int32_t  kint[K_MAX_CHANNELS][LPC_MAX_ORDER + 1];
  uint32_t rshift[K_MAX_CHANNELS], pitch[K_MAX_CHANNELS], rice_init[K_MAX_CHANNELS];
  int32_t  ltm_q[K_MAX_CHANNELS][LU_MAX];
  int32_t* lattice_keep[K_MAX_CHANNELS];   /* residual right after the lattice */
  uint32_t block_type;
  int      skip_pack;               /* hot-path timing: stop before the bit-serial data body */
  double   r_scratch[LPC_MAX_ORDER + 1];   /* the LPC calculator's persistent autocorrelation buffer */
} enc_t;

static void enc_free(enc_t* e)
{
  uint32_t ch;
  for (ch = 0; ch < K_MAX_CHANNELS; ch++) {
    free(e->xd[ch]); free(e->xi[ch]); free(e->res[ch]); free(e->tmp[ch]); free(e->lattice_keep[ch]);
  }
  free(e->window);
  free(e);
}

/* capacity checks follow SLAEncoder_SetWaveFormat / SetEncodeParameter (src/SLAEncoder.c:176-224) */
static enc_t* enc_new(const sla_flat_params* p, int* err)
{
  enc_t* e;
  uint32_t ch;
  *err = SLAO_OK;
  if (p == NULL) { *err = SLAO_INVALID_ARGUMENT; return NULL; }
  if (p->num_channels > p->cap_channels || p->bits_per_sample > 32 || p->cap_channels > K_MAX_CHANNELS
      || p->num_channels == 0) { *err = SLAO_EXCEED_HANDLE_CAPACITY; return NULL; }
  if (p->parcor_order > p->cap_parcor_order || p->longterm_order > p->cap_longterm_order
      || p->lms_order > p->cap_lms_order || p->max_block_samples > p->cap_block_samples
      || p->max_block_samples < K_MIN_BLOCK || p->cap_parcor_order > LPC_MAX_ORDER
      || p->cap_longterm_order > LU_MAX) { *err = SLAO_EXCEED_HANDLE_CAPACITY; return NULL; }
  e = (enc_t*)calloc(1, sizeof(enc_t));
  e->p = *p;
  e->fft_size = pow2ceil32(p->cap_block_samples * 2);
  for (ch = 0; ch < p->cap_channels; ch++) {
    e->xd[ch]  = (double*)malloc(sizeof(double) * p->cap_block_samples);
    e->xi[ch]  = (int32_t*)malloc(sizeof(int32_t) * p->cap_block_samples);
    e->res[ch] = (int32_t*)calloc(p->cap_block_samples, sizeof(int32_t));
    e->tmp[ch] = (int32_t*)malloc(sizeof(int32_t) * p->cap_block_samples);
    e->lattice_keep[ch] = (int32_t*)malloc(sizeof(int32_t) * p->cap_block_samples);
  }
  e->window = (double*)malloc(sizeof(double) * p->cap_block_samples);
  e->window_n = 0;
  return e;
}

static void put_be16(uint8_t* d, uint32_t v) { d[0] = (uint8_t)(v >> 8); d[1] = (uint8_t)v; }
static void put_be32(uint8_t* d, uint32_t v) { d[0] = (uint8_t)(v >> 24); d[1] = (uint8_t)(v >> 16); d[2] = (uint8_t)(v >> 8); d[3] = (uint8_t)v; }
static uint32_t get_be16(const uint8_t* d) { return ((uint32_t)d[0] << 8) | d[1]; }
static uint32_t get_be32(const uint8_t* d) { return ((uint32_t)d[0] << 24) | ((uint32_t)d[1] << 16) | ((uint32_t)d[2] << 8) | d[3]; }

/* 43-byte big-endian file header (reference src/SLAEncoder.c:243-289) */
static int write_header(const sla_flat_params* p, uint32_t lshift, uint32_t num_samples, uint32_t num_blocks,
                        uint32_t max_block_size, uint32_t max_bps, uint8_t* d, uint32_t cap)
{
  if (d == NULL) { return SLAO_INVALID_ARGUMENT; }
  if (cap < K_HEADER_SIZE) { return SLAO_INSUFFICIENT_BUFFER_SIZE; }
  d[0] = 'S'; d[1] = 'L'; d[2] = '*'; d[3] = 1;
  put_be32(d + 4, K_HEADER_SIZE - 8);
  put_be16(d + 8, 0);
  put_be32(d + 10, 1);
  d[14] = (uint8_t)p->num_channels;
  put_be32(d + 15, num_samples);
  put_be32(d + 19, p->sampling_rate);
  d[23] = (uint8_t)p->bits_per_sample;
  d[24] = (uint8_t)lshift;
  d[25] = (uint8_t)p->parcor_order;
  d[26] = (uint8_t)p->longterm_order;
  d[27] = (uint8_t)p->lms_order;
  d[28] = (uint8_t)p->ch_process_method;
  put_be32(d + 29, num_blocks);
  put_be16(d + 33, p->max_block_samples);
  put_be32(d + 35, max_block_size);
  put_be32(d + 39, max_bps);
  put_be16(d + 8, slao_crc16(d + K_HDR_CRC_START, K_HEADER_SIZE - K_HDR_CRC_START));
  return SLAO_OK;
}

/* A0 staging: double scaling, integer right-justify, optional mid/side
 * (reference src/SLAEncoder.c:505-515, 381-390; src/SLAUtility.c:370-412) */
static int stage_input(enc_t* e, const int32_t* const* in, uint32_t n, uint32_t shift)
{
  const uint32_t C = e->p.num_channels;
  uint32_t ch, s;
  for (ch = 0; ch < C; ch++) {
    for (s = 0; s < n; s++) {
      e->xd[ch][s] = (double)in[ch][s] * ldexp(1.0, -31);
      e->xi[ch][s] = in[ch][s] >> shift;
    }
  }
  if (e->p.ch_process_method == 1) {
    if (C != 2) { return SLAO_INVALID_CHPROCESSMETHOD; }
    for (s = 0; s < n; s++) {
      double l = e->xd[0][s], r = e->xd[1][s];
      int32_t li = e->xi[0][s], ri = e->xi[1][s];
      e->xd[0][s] = (l + r) / 2;
      e->xd[1][s] = l - r;
      e->xi[0][s] = add_wrap(li, ri) >> 1;
      e->xi[1][s] = sub_wrap(li, ri);
    }
  }
  return SLAO_OK;
}

/* one block (reference src/SLAEncoder.c:458-801) */
static int encode_block(enc_t* e, const int32_t* const* in, uint32_t n, uint8_t* out, uint32_t cap, uint32_t* out_size)
{
  const sla_flat_params* p = &e->p;
  const uint32_t C = p->num_channels, order = p->parcor_order, ntaps = p->longterm_order;
  uint32_t ch, s, ord;
  bitw_t w;
  int ret;

  if (in == NULL || out == NULL || out_size == NULL) { return SLAO_INVALID_ARGUMENT; }
  if (n > p->cap_block_samples) { return SLAO_EXCEED_HANDLE_CAPACITY; }
  if (cap <= K_BLOCK_HEADER_MIN) { return SLAO_INSUFFICIENT_DATA_SIZE; }
  if (p->window_type > 4) { return SLAO_INVALID_WINDOWFUNCTION_TYPE; }
  if (e->window_n != n || e->window_type != p->window_type) {
    slao_window(p->window_type, e->window, n);
    e->window_n = n; e->window_type = p->window_type;
  }
  if ((ret = stage_input(e, in, n, 32 - p->bits_per_sample + e->lshift)) != SLAO_OK) { return ret; }

  e->block_type = BLK_SILENT;
  for (ch = 0; ch < C && e->block_type == BLK_SILENT; ch++) {
    for (s = 0; s < n; s++) { if (e->xi[ch][s] != 0) { e->block_type = BLK_COMPRESS; break; } }
  }

  for (ch = 0; ch < C && e->block_type == BLK_COMPRESS; ch++) {
    double* xd = e->xd[ch];
    double est, r0 = 0.0;
    uint32_t bw;
    int lret;
    for (s = 0; s < n; s++) { xd[s] *= e->window[s]; }
    slao_preemph_f64(xd, n);
    if (parcor_with_scratch(e->r_scratch, xd, n, order, e->parcor[ch]) != 0) { return SLAO_FAILED_TO_CALCULATE_COEF; }
    for (s = 0; s < n; s++) { r0 += xd[s] * xd[s]; }
    est = code_length_from_power(r0, n, p->bits_per_sample, e->parcor[ch], order);
    est = (8 * est) / p->bits_per_sample;
    if (est >= K_RAW_THRESHOLD) { e->block_type = BLK_RAW; break; }

    /* A6 quantiser (src/SLAEncoder.c:567-589) */
    bw = slao_bitwidth(e->xi[ch], n);
    e->rshift[ch] = (bw > 16) ? (bw - 16) : 0;
    e->kint[ch][0] = 0; e->code[ch][0] = 0;
    for (ord = 1; ord <= order; ord++) {
      uint32_t q = (ord < 4) ? 16 : 8;
      int32_t lim = 1 << (q - 1);
      int32_t c = f64_to_i32_x86(round_half_away(e->parcor[ch][ord] * ldexp(1.0, (int)q - 1)));
      if (c < -lim) { c = -lim; }
      if (c > lim - 1) { c = lim - 1; }
      e->code[ch][ord] = c;
      e->kint[ch][ord] = shl_wrap(c, 16u - q) >> e->rshift[ch];
    }

    memcpy(e->tmp[ch], e->xi[ch], sizeof(int32_t) * n);
    slao_preemph_i32(e->tmp[ch], n);
    slao_lattice_predict(e->tmp[ch], n, e->kint[ch], order, e->res[ch]);
    memcpy(e->lattice_keep[ch], e->res[ch], sizeof(int32_t) * n);

    /* long-term stage (src/SLAEncoder.c:619-656) */
    {
      double coef[LU_MAX];
      for (ord = 0; ord < LU_MAX; ord++) { coef[ord] = 0.0; }
      lret = slao_ltm_analyze(e->res[ch], n, e->fft_size, p->cap_longterm_order, ntaps, &e->pitch[ch], coef, NULL);
      if (lret != 0 && lret != 4) { return SLAO_FAILED_TO_CALCULATE_COEF; }
      if (lret == 4 || e->pitch[ch] >= K_LTM_MAX_PERIOD) { e->pitch[ch] = 0; }
      /* on a failed analysis the reference quantises whatever its coefficient
       * buffer held before; those values never reach the stream (pitch = 0) */
      for (ord = 0; ord < ntaps; ord++) {
        e->ltm_q[ch][ord] = shl_wrap(f64_to_i32_x86(round_half_away(coef[ord] * ldexp(1.0, 15))), 16);
      }
      if (e->pitch[ch] >= K_LTM_MIN_PITCH) {
        slao_ltm_predict(e->res[ch], n, e->pitch[ch], e->ltm_q[ch], ntaps, e->tmp[ch]);
        memcpy(e->res[ch], e->tmp[ch], sizeof(int32_t) * n);
      }
    }
    /* LMS stage (src/SLAEncoder.c:658-670) */
    if (slao_lms_predict(e->res[ch], n, p->lms_order, e->tmp[ch]) != 0) { return SLAO_FAILED_TO_PREDICT; }
    memcpy(e->res[ch], e->tmp[ch], sizeof(int32_t) * n);
  }

  if (e->block_type == BLK_COMPRESS) {
    for (ch = 0; ch < C; ch++) { e->rice_init[ch] = rice_init_1ch(e->res[ch], n); }
  }

  /* block header + parameters (src/SLAEncoder.c:682-737) */
  bw_open(&w, out, cap);
  bw_put(&w, K_SYNC, 16);
  bw_put(&w, 0, 32);
  bw_put(&w, 0, 16);
  bw_put(&w, n, 16);
  bw_put(&w, e->block_type, 2);
  if (e->block_type == BLK_COMPRESS) {
    for (ch = 0; ch < C; ch++) {
      bw_put(&w, e->rshift[ch], 4);
      for (ord = 1; ord <= order; ord++) { bw_put(&w, fold32(e->code[ch][ord]), (ord < 4) ? 16 : 8); }
      if (e->pitch[ch] >= K_LTM_MIN_PITCH) {
        bw_put(&w, 1, 1);
        bw_put(&w, e->pitch[ch], K_LTM_PERIOD_BITS);
        for (ord = 0; ord < ntaps; ord++) { bw_put(&w, fold32(e->ltm_q[ch][ord] >> 16), 16); }
      } else {
        bw_put(&w, 0, 1);
      }
      bw_put(&w, rp_get(rp_set(e->rice_init[ch])), p->bits_per_sample);
    }
  }
  bw_align(&w);

  /* body (src/SLAEncoder.c:740-778) */
  if (e->block_type == BLK_RAW) {
    uint32_t nbits[K_MAX_CHANNELS];
    for (ch = 0; ch < C; ch++) {
      nbits[ch] = p->bits_per_sample - e->lshift;
      if (ch == 1 && p->ch_process_method == 1) { nbits[ch] += 1; }
    }
    for (s = 0; s < n; s++) { for (ch = 0; ch < C; ch++) { bw_put(&w, fold32(e->xi[ch][s]), nbits[ch]); } }
  } else if (e->block_type == BLK_COMPRESS && !e->skip_pack) {
    put_residuals(&w, (const int32_t* const*)e->res, C, n, e->rice_init);
  }
  bw_align(&w);
  *out_size = (uint32_t)bw_tell(&w);
  put_be32(out + 2, *out_size - 6);
  put_be16(out + 6, slao_crc16(out + K_BLK_CRC_START, *out_size - K_BLK_CRC_START));
  return w.overflow ? SLAO_INSUFFICIENT_DATA_SIZE : SLAO_OK;
}

/* super-frame: leading-silence shortcut, else partition search
 * (reference src/SLAEncoder.c:356-422) */
static int plan_superframe(enc_t* e, const int32_t* const* in, uint32_t n, uint32_t min_blk,
                           uint32_t* nparts, uint32_t* parts)
{
  const sla_flat_params* p = &e->p;
  const uint32_t C = p->num_channels;
  uint32_t s, ch;
  double* flat;
  int ret;
  if (n < min_blk) { return SLAO_INVALID_ARGUMENT; }
  if ((ret = stage_input(e, in, n, 32 - p->bits_per_sample)) != SLAO_OK) { return ret; }
  for (s = 0; s < n; s++) {
    int nz = 0;
    for (ch = 0; ch < C; ch++) { if (e->xi[ch][s] != 0) { nz = 1; break; } }
    if (nz) { break; }
  }
  if (s >= min_blk) { *nparts = 1; parts[0] = s; return SLAO_OK; }
  flat = (double*)malloc(sizeof(double) * (size_t)C * n);
  for (ch = 0; ch < C; ch++) { memcpy(&flat[(size_t)ch * n], e->xd[ch], sizeof(double) * n); }
  ret = partition_search_scratch(e->r_scratch, flat, C, n, min_blk, K_SEARCH_DELTA, n, p->bits_per_sample, p->parcor_order, nparts, parts);
  free(flat);
  return (ret == 0) ? SLAO_OK : SLAO_FAILED_TO_CALCULATE_COEF;
}

/* offset_lshift: trailing zero bits common to every sample
 * (reference src/SLAEncoder.c:425-455) */
static uint32_t common_lshift(const sla_flat_params* p, const int32_t* const* in, uint32_t n)
{
  uint32_t mask = 0, ch, s, used;
  for (ch = 0; ch < p->num_channels; ch++) { for (s = 0; s < n; s++) { mask |= (uint32_t)in[ch][s]; } }
  if (mask == 0) { return 0; }
  used = 1 + log2floor32(~mask & (mask - 1u));   /* = number of trailing zeros... see below */
  return p->bits_per_sample - (32 - used);
}

static void trace_block(enc_t* e, sla_flat_trace* tr, uint32_t b, uint32_t pos, uint32_t n, uint32_t bytes)
{
  const uint32_t C = e->p.num_channels, O = e->p.parcor_order + 1;
  uint32_t ch, ord;
  if (tr == NULL || b >= tr->max_blocks) { return; }
  tr->blk_start[b] = pos; tr->blk_nsmpl[b] = n; tr->blk_type[b] = e->block_type; tr->blk_bytes[b] = bytes;
  for (ch = 0; ch < C; ch++) {
    size_t bc = (size_t)b * C + ch;
    if (e->block_type != BLK_COMPRESS) { tr->rshift[bc] = 0; tr->pitch[bc] = 0; tr->rice_init[bc] = 0; continue; }
    for (ord = 0; ord < O; ord++) {
      tr->parcor[bc * tr->order_stride + ord] = e->parcor[ch][ord];
      tr->code[bc * tr->order_stride + ord] = e->code[ch][ord];
      tr->kint[bc * tr->order_stride + ord] = e->kint[ch][ord];
    }
    tr->rshift[bc] = e->rshift[ch];
    tr->pitch[bc] = e->pitch[ch];
    for (ord = 0; ord < e->p.longterm_order; ord++) { tr->ltm_coef[bc * tr->ltm_stride + ord] = e->ltm_q[ch][ord]; }
    tr->rice_init[bc] = rp_get(rp_set(e->rice_init[ch]));
    memcpy(&tr->res_final[(size_t)ch * tr->sample_stride + pos], e->res[ch], sizeof(int32_t) * n);
    memcpy(&tr->res_lattice[(size_t)ch * tr->sample_stride + pos], e->lattice_keep[ch], sizeof(int32_t) * n);
  }
}

/* whole file (reference src/SLAEncoder.c:804-932) */
/* forced_lshift != 0xFFFFFFFF: the samples are a range of a longer file whose offset_lshift this is (tests of the
 * multi-GPU sharding: the reference computes it over the whole input, src/SLAEncoder.c:835-837) */
static int encode_whole_impl(const sla_flat_params* p, const int32_t* input, uint32_t n, uint8_t* out, uint32_t cap,
                             uint32_t* out_size, sla_flat_trace* tr, int skip_pack, uint32_t forced_lshift)
{
  const int32_t* chan[K_MAX_CHANNELS];
  const int32_t* at[K_MAX_CHANNELS];
  uint32_t parts[MAX_NODES], nparts, ch, pos = 0, cur = K_HEADER_SIZE, nblocks = 0, maxblk = 0, maxbps = 0;
  enc_t* e;
  int ret;

  if (input == NULL || out == NULL || out_size == NULL) { return SLAO_INVALID_ARGUMENT; }
  e = enc_new(p, &ret);
  if (e == NULL) { return ret; }
  e->skip_pack = skip_pack;
  for (ch = 0; ch < p->num_channels; ch++) { chan[ch] = &input[(size_t)ch * n]; }
  if ((ret = write_header(p, 0, n, 0, 0xFFFFFFFFu, 0, out, cap)) != SLAO_OK) { goto done; }
  e->lshift = (forced_lshift != 0xFFFFFFFFu) ? forced_lshift : common_lshift(p, chan, n);
  if (tr != NULL) { tr->offset_lshift = e->lshift; }

  while (pos < n) {
    uint32_t remain = n - pos, win = (p->max_block_samples < remain) ? p->max_block_samples : remain, part;
    if (cur >= cap) { ret = SLAO_INSUFFICIENT_BUFFER_SIZE; goto done; }
    for (ch = 0; ch < p->num_channels; ch++) { at[ch] = chan[ch] + pos; }
    ret = plan_superframe(e, at, win, (K_MIN_BLOCK < remain) ? K_MIN_BLOCK : remain, &nparts, parts);
    if (ret != SLAO_OK) { goto done; }
    for (part = 0; part < nparts; part++) {
      uint32_t cnt = parts[part], bsize, bps_blk;
      for (ch = 0; ch < p->num_channels; ch++) { at[ch] = chan[ch] + pos; }
      if ((ret = encode_block(e, at, cnt, out + cur, cap - cur, &bsize)) != SLAO_OK) { goto done; }
      trace_block(e, tr, nblocks, pos, cnt, bsize);
      cur += bsize; pos += cnt;
      if (bsize > maxblk) { maxblk = bsize; }
      bps_blk = (8 * bsize * p->sampling_rate) / cnt;
      if (bps_blk > maxbps) { maxbps = bps_blk; }
      nblocks++;
    }
  }
  if (cur > cap) { ret = SLAO_INSUFFICIENT_DATA_SIZE; goto done; }
  ret = write_header(p, e->lshift, n, nblocks, maxblk, maxbps, out, cap);
  *out_size = cur;
  if (tr != NULL) { tr->num_blocks = nblocks; }
done:
  enc_free(e);
  return ret;
}

int slao_encode_whole(const sla_flat_params* p, const int32_t* input, uint32_t n, uint8_t* out, uint32_t cap, uint32_t* out_size)
{ return encode_whole_impl(p, input, n, out, cap, out_size, NULL, 0, 0xFFFFFFFFu); }

int slao_encode_range(const sla_flat_params* p, const int32_t* input, uint32_t n, uint32_t file_lshift, uint8_t* out, uint32_t cap, uint32_t* out_size)
{ return encode_whole_impl(p, input, n, out, cap, out_size, NULL, 0, file_lshift); }

int slao_encode_trace(const sla_flat_params* p, const int32_t* input, uint32_t n, uint8_t* out, uint32_t cap,
                      uint32_t* out_size, sla_flat_trace* tr)
{ return encode_whole_impl(p, input, n, out, cap, out_size, tr, 0, 0xFFFFFFFFu); }

/* CPU-baseline leg of bench.py: the LPC+residual path only (everything up to and including the
 * Rice initial parameter; the bit-serial residual body is not emitted) */
int slao_hotpath(const sla_flat_params* p, const int32_t* input, uint32_t n, uint8_t* out, uint32_t cap,
                 uint32_t* out_size, sla_flat_trace* tr)
{ return encode_whole_impl(p, input, n, out, cap, out_size, tr, 1, 0xFFFFFFFFu); }

/* fixed-size EncodeBlock calls under one header */
int slao_encode_fixed_blocks(const sla_flat_params* p, const int32_t* input, uint32_t n, uint32_t block_samples,
                             uint8_t* out, uint32_t cap, uint32_t* out_size)
{
  const int32_t* at[K_MAX_CHANNELS];
  uint32_t ch, pos, cur = K_HEADER_SIZE, nblocks = 0, maxblk = 0, maxbps = 0;
  enc_t* e;
  int ret;
  if (input == NULL || out == NULL || out_size == NULL || block_samples == 0) { return SLAO_INVALID_ARGUMENT; }
  e = enc_new(p, &ret);
  if (e == NULL) { return ret; }
  if (cap < K_HEADER_SIZE) { enc_free(e); return SLAO_INSUFFICIENT_BUFFER_SIZE; }
  for (pos = 0; pos < n; pos += block_samples) {
    uint32_t cnt = (n - pos < block_samples) ? (n - pos) : block_samples, bsize, bps_blk;
    for (ch = 0; ch < p->num_channels; ch++) { at[ch] = &input[(size_t)ch * n + pos]; }
    if ((ret = encode_block(e, at, cnt, out + cur, cap - cur, &bsize)) != SLAO_OK) { enc_free(e); return ret; }
    cur += bsize;
    if (bsize > maxblk) { maxblk = bsize; }
    bps_blk = (8 * bsize * p->sampling_rate) / cnt;
    if (bps_blk > maxbps) { maxbps = bps_blk; }
    nblocks++;
  }
  ret = write_header(p, 0, n, nblocks, maxblk, maxbps, out, cap);
  *out_size = cur;
  enc_free(e);
  return ret;
}

/* ======================================================================== */
/* decoder (reference src/SLADecoder.c:157-254, 309-566, 660-732)            */
/* ======================================================================== */
int slao_decode_whole(const sla_flat_params* p, const uint8_t* data, uint32_t size, int32_t* out, uint32_t nmax,
                      uint32_t* nsamples, uint32_t* hdr_out)
{
  uint32_t C, bps, lshift, order, ntaps, lms, ms, total, pos = 0, off = K_HEADER_SIZE, ch, s, ord;
  int32_t *res[K_MAX_CHANNELS], *buf[K_MAX_CHANNELS];
  int ret = SLAO_OK, hdr_status = SLAO_OK;
  uint32_t capn;

  if (data == NULL || out == NULL || nsamples == NULL || p == NULL) { return SLAO_INVALID_ARGUMENT; }
  if (size < K_HEADER_SIZE) { return SLAO_INSUFFICIENT_DATA_SIZE; }
  if (data[0] != 'S' || data[1] != 'L' || data[2] != '*' || data[3] != 1) { return SLAO_INVALID_HEADER_FORMAT; }
  if (get_be16(data + 8) != slao_crc16(data + K_HDR_CRC_START, K_HEADER_SIZE - K_HDR_CRC_START)) { hdr_status = SLAO_DETECT_DATA_CORRUPTION; }
  if (get_be32(data + 10) != 1) { return SLAO_INVALID_HEADER_FORMAT; }
  C = data[14]; total = get_be32(data + 15); bps = data[23]; lshift = data[24];
  order = data[25]; ntaps = data[26]; lms = data[27]; ms = data[28];
  if (hdr_out != NULL) {
    hdr_out[0] = C; hdr_out[1] = bps; hdr_out[2] = get_be32(data + 19); hdr_out[3] = lshift;
    hdr_out[4] = order; hdr_out[5] = ntaps; hdr_out[6] = lms; hdr_out[7] = ms;
    hdr_out[8] = total; hdr_out[9] = get_be32(data + 29); hdr_out[10] = get_be16(data + 33); hdr_out[11] = get_be32(data + 35);
  }
  if (hdr_status != SLAO_OK) { return hdr_status; }
  if (C > p->cap_channels || C == 0 || C > K_MAX_CHANNELS || bps > 32) { return SLAO_EXCEED_HANDLE_CAPACITY; }
  if (order > p->cap_parcor_order || ntaps > p->cap_longterm_order || lms > p->cap_lms_order
      || get_be16(data + 33) > p->cap_block_samples || get_be16(data + 33) < K_MIN_BLOCK
      || order > LPC_MAX_ORDER || ntaps > LU_MAX) { return SLAO_EXCEED_HANDLE_CAPACITY; }
  if (ms == 1 && C != 2) { return SLAO_INVALID_CHPROCESSMETHOD; }
  capn = p->cap_block_samples;
  for (ch = 0; ch < C; ch++) {
    res[ch] = (int32_t*)malloc(sizeof(int32_t) * capn);
    buf[ch] = (int32_t*)malloc(sizeof(int32_t) * capn);
  }

  while (pos < total) {
    bitr_t r;
    uint32_t bsize, n, type, init[K_MAX_CHANNELS], rsh, pitch[K_MAX_CHANNELS];
    int32_t kint[K_MAX_CHANNELS][LPC_MAX_ORDER + 1], ltm[K_MAX_CHANNELS][LU_MAX];
    const uint8_t* blk = data + off;
    uint32_t left;
    if (off > size) { ret = SLAO_INSUFFICIENT_DATA_SIZE; break; }
    left = size - off;
    if (left < 11) { ret = SLAO_INSUFFICIENT_DATA_SIZE; break; }
    br_open(&r, blk, left);
    if (br_get(&r, 16) != K_SYNC) { ret = SLAO_FAILED_TO_FIND_SYNC_CODE; break; }
    bsize = br_get(&r, 32) + 6;
    {
      uint32_t crc = br_get(&r, 16);
      if (left >= bsize && bsize >= K_BLK_CRC_START && slao_crc16(blk + K_BLK_CRC_START, bsize - K_BLK_CRC_START) != crc) {
        ret = SLAO_DETECT_DATA_CORRUPTION; break;
      }
    }
    n = br_get(&r, 16);
    type = br_get(&r, 2);
    if (bsize > left) { ret = SLAO_INSUFFICIENT_DATA_SIZE; break; }
    if (n > nmax - pos || n > capn) { ret = SLAO_INSUFFICIENT_BUFFER_SIZE; break; }
    if (type == BLK_COMPRESS) {
      for (ch = 0; ch < C; ch++) {
        rsh = br_get(&r, 4);
        kint[ch][0] = 0;
        for (ord = 1; ord <= order; ord++) {
          uint32_t q = (ord < 4) ? 16 : 8;
          kint[ch][ord] = shl_wrap(unfold32(br_get(&r, q)), 16u - q) >> rsh;
        }
        if (br_get(&r, 1) == 0) { pitch[ch] = 0; }
        else {
          pitch[ch] = br_get(&r, K_LTM_PERIOD_BITS);
          for (ord = 0; ord < ntaps; ord++) { ltm[ch][ord] = shl_wrap(unfold32(br_get(&r, 16)), 16); }
        }
        init[ch] = br_get(&r, bps);
      }
    }
    br_align(&r);
    if (type == BLK_SILENT) {
      for (ch = 0; ch < C; ch++) { memset(buf[ch], 0, sizeof(int32_t) * n); }
    } else if (type == BLK_RAW) {
      for (s = 0; s < n; s++) {
        for (ch = 0; ch < C; ch++) {
          uint32_t nb = bps - lshift + ((ch == 1 && ms == 1) ? 1 : 0);
          buf[ch][s] = unfold32(br_get(&r, nb));
        }
      }
    } else if (type == BLK_COMPRESS) {
      get_residuals(&r, res, C, n, init);
      for (ch = 0; ch < C; ch++) {
        if (lms_run(res[ch], n, lms, buf[ch], 0) != 0) { ret = SLAO_FAILED_TO_SYNTHESIZE; break; }
        memcpy(res[ch], buf[ch], sizeof(int32_t) * n);
        if (pitch[ch] != 0) {
          ltm_run(res[ch], n, pitch[ch], ltm[ch], ntaps, buf[ch], 0);
          memcpy(res[ch], buf[ch], sizeof(int32_t) * n);
        }
        slao_lattice_synth(res[ch], n, kint[ch], order, buf[ch]);
        slao_deemph_i32(buf[ch], n);
      }
      if (ret != SLAO_OK) { break; }
    } else { ret = SLAO_INVALID_HEADER_FORMAT; break; }
    if (ms == 1) {
      for (s = 0; s < n; s++) {
        int32_t side = buf[1][s], mid = (int32_t)(((uint32_t)buf[0][s] << 1) | ((uint32_t)side & 1u));
        buf[0][s] = add_wrap(mid, side) >> 1;
        buf[1][s] = sub_wrap(mid, side) >> 1;
      }
    }
    for (ch = 0; ch < C; ch++) {
      for (s = 0; s < n; s++) { out[(size_t)ch * nmax + pos + s] = shl_wrap(buf[ch][s], 32 - bps + lshift); }
    }
    br_align(&r);
    off += (uint32_t)br_tell(&r);
    pos += n;
  }
  *nsamples = pos;
  for (ch = 0; ch < C; ch++) { free(res[ch]); free(buf[ch]); }
  return ret;
}
