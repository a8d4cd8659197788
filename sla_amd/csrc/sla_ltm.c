/*
 * sla_ltm.c -- long-term (pitch) predictor analysis, host side.
 *
 * The reference derives the pitch and the 1/3/5 taps from an FFT-based
 * autocorrelation of the lattice residual (src/SLAPredictor.c:791-980) using
 * a Numerical-Recipes style real FFT whose twiddles come from a trigonometric
 * recurrence seeded by libm sin() (src/SLAUtility.c:220-312).  To stay
 * bit-exact the twiddle sequences are generated here, once per FFT size, by
 * the same recurrence in host double arithmetic (slai_fft_plan); the
 * butterflies then only multiply/add doubles in a fixed order.
 *
 * Only the tiny Toeplitz solve (x87 long double residual, src/SLAUtility.c:627-657)
 * stays on the host: a few hundred flops per block; the peak scan runs in k_ltm_acf.
 */
#include "sla_internal.h"

#include <float.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>

struct slai_fft_plan {
  uint32_t fft_size;        /* real length n (power of two) */
  /* complex FFT of n/2 points: for stage with half-span `mmax` (2,4,..,n/2) the twiddles of
   * butterfly group k=0..mmax/2-1 are at tw[dir][mmax/2 - 1 + k] (dir 0 forward, 1 inverse) */
  double* tw_re[2];
  double* tw_im[2];
  /* real post/pre-processing twiddles for i = 2 .. n/4 at index i-2 */
  double* rt_re[2];
  double* rt_im[2];
};

static void fill_stage_twiddles(double* re, double* im, uint32_t n, int isign)
{
  uint32_t mmax;
  for (mmax = 2; mmax < n; mmax <<= 1) {
    const double theta = isign * (6.28318530717959 / (double)mmax);
    double wtemp = sin(0.5 * theta);
    const double wpr = -2.0 * wtemp * wtemp;
    const double wpi = sin(theta);
    double wr = 1.0, wi = 0.0;
    uint32_t k, base = mmax / 2 - 1;
    for (k = 0; k < mmax / 2; k++) {
      re[base + k] = wr; im[base + k] = wi;
      wtemp = wr;
      wr = wtemp * wpr - wi * wpi + wr;
      wi = wi * wpr + wtemp * wpi + wi;
    }
  }
}

static void fill_real_twiddles(double* re, double* im, uint32_t n, int isign)
{
  double theta = 3.141592653589793 / (double)(n >> 1);
  double wtemp, wpr, wpi, wr, wi;
  uint32_t i;
  if (isign != 1) { theta = -theta; }
  wtemp = sin(0.5 * theta);
  wpr = -2.0 * wtemp * wtemp;
  wpi = sin(theta);
  wr = 1.0 + wpr;
  wi = wpi;
  for (i = 2; i <= (n >> 2); i++) {
    re[i - 2] = wr; im[i - 2] = wi;
    wtemp = wr;
    wr = wtemp * wpr - wi * wpi + wr;
    wi = wi * wpr + wtemp * wpi + wi;
  }
}

slai_fft_plan* slai_fft_plan_create(uint32_t fft_size)
{
  slai_fft_plan* p;
  int d;
  if (fft_size < 8 || (fft_size & (fft_size - 1))) { return NULL; }
  p = (slai_fft_plan*)calloc(1, sizeof(*p));
  if (p == NULL) { return NULL; }
  p->fft_size = fft_size;
  for (d = 0; d < 2; d++) {
    p->tw_re[d] = (double*)malloc(sizeof(double) * (fft_size / 2));
    p->tw_im[d] = (double*)malloc(sizeof(double) * (fft_size / 2));
    p->rt_re[d] = (double*)malloc(sizeof(double) * (fft_size / 4));
    p->rt_im[d] = (double*)malloc(sizeof(double) * (fft_size / 4));
    fill_stage_twiddles(p->tw_re[d], p->tw_im[d], fft_size, d == 0 ? 1 : -1);
    fill_real_twiddles(p->rt_re[d], p->rt_im[d], fft_size, d == 0 ? 1 : -1);
  }
  return p;
}

void slai_fft_plan_destroy(slai_fft_plan* p)
{
  int d;
  if (p == NULL) { return; }
  for (d = 0; d < 2; d++) { free(p->tw_re[d]); free(p->tw_im[d]); free(p->rt_re[d]); free(p->rt_im[d]); }
  free(p);
}

uint32_t slai_fft_plan_size(const slai_fft_plan* p) { return p->fft_size; }

/* flat copy for the device kernels (layout documented at k_ltm_acf in sla_kernels.hip): SLA_HIP_TWIDDLE_DOUBLES(fft_size)
 * = 6 * fft_size doubles -- the split re / im tables of k_ltm_acf, then the same values as (re, im) pairs for k_ltm_acf2,
 * which fetches a twiddle with one 16-byte load */
void slai_fft_plan_export(const slai_fft_plan* p, double* out)
{
  const uint32_t F = p->fft_size, half = F / 2, quarter = F / 4;
  double* pair = out + 3 * (size_t)F;
  uint32_t i, d;
  memset(out, 0, sizeof(double) * 6 * (size_t)F);
  memcpy(out,                       p->tw_re[0], sizeof(double) * (half - 1));
  memcpy(out + half,                p->tw_im[0], sizeof(double) * (half - 1));
  memcpy(out + F,                   p->tw_re[1], sizeof(double) * (half - 1));
  memcpy(out + F + half,            p->tw_im[1], sizeof(double) * (half - 1));
  memcpy(out + 2 * F,               p->rt_re[0], sizeof(double) * (quarter - 1));
  memcpy(out + 2 * F + quarter,     p->rt_im[0], sizeof(double) * (quarter - 1));
  memcpy(out + 2 * F + 2 * quarter, p->rt_re[1], sizeof(double) * (quarter - 1));
  memcpy(out + 2 * F + 3 * quarter, p->rt_im[1], sizeof(double) * (quarter - 1));
  /* pairs: [0, F/2) forward stages | [F/2, F) inverse stages | [F, F + F/4) forward recombination | [F + F/4, F + F/2) inverse */
  for (d = 0; d < 2; d++) {
    for (i = 0; i + 1 < half; i++) { pair[2 * ((size_t)d * half + i)] = p->tw_re[d][i]; pair[2 * ((size_t)d * half + i) + 1] = p->tw_im[d][i]; }
    for (i = 0; i + 1 < quarter; i++) { pair[2 * ((size_t)F + (size_t)d * quarter + i)] = p->rt_re[d][i]; pair[2 * ((size_t)F + (size_t)d * quarter + i) + 1] = p->rt_im[d][i]; }
  }
}

/* ---- tiny dense solve (reference src/SLAUtility.c:487-674) ---------------- */
#define NT SLAI_MAX_TAPS
static int lu_factor(double A[NT][NT], uint32_t dim, uint32_t* perm, double* scale)
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
      for (k = 0; k < dim; k++) { const double t = A[imax][k]; A[imax][k] = A[col][k]; A[col][k] = t; }
      scale[imax] = scale[col];
    }
    perm[col] = imax;
    if (fabs(A[col][col]) <= FLT_EPSILON) { return -1; }
    if (col != dim - 1) {
      const double inv = 1.0f / A[col][col];
      for (row = col + 1; row < dim; row++) { A[row][col] *= inv; }
    }
  }
  return 0;
}

static void lu_substitute(double A[NT][NT], double* b, uint32_t dim, const uint32_t* perm)
{
  uint32_t row, col, first_nz = 0;
  double sum;
  for (row = 0; row < dim; row++) {
    const uint32_t pv = perm[row];
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

static int toeplitz_solve(const double R[NT][NT], double* b, uint32_t dim, uint32_t refinements)
{
  double A[NT][NT], x[NT], err[NT], scale[NT];
  uint32_t perm[NT], row, col, it;
  memcpy(A, R, sizeof(A));
  memcpy(x, b, sizeof(double) * dim);
  if (lu_factor(A, dim, perm, scale) != 0) { return -1; }
  lu_substitute(A, x, dim, perm);
  for (it = 0; it < refinements; it++) {
    for (row = 0; row < dim; row++) {
      long double e = -b[row];                      /* extended precision, as the reference */
      for (col = 0; col < dim; col++) { e += R[row][col] * x[col]; }
      err[row] = (double)e;
    }
    lu_substitute(A, err, dim, perm);
    for (row = 0; row < dim; row++) { x[row] -= err[row]; }
  }
  memcpy(b, x, sizeof(double) * dim);
  return 0;
}

/* Long-term taps from the device's compact record {code, chosen lag, acf[0..4], acf[chosen-2..chosen+2]}
 * (k_ltm_acf did the FFT autocorrelation and the peak scan): range check, Wiener solution around the
 * chosen lag, stability fallback (reference src/SLAPredictor.c:855-863, 913-979).
 * Returns 0 ok, 4 analysis failed (SLAPREDICTOR_APIRESULT_FAILED_TO_CALCULATION). */
int slai_ltm_solve(const double* rec, uint32_t ntaps, uint32_t* pitch, double* coef)
{
  const double* low = rec + 2;            /* acf[0..4]                 */
  const double* mid = rec + 7;            /* acf[chosen-2..chosen+2]   */
  const uint32_t chosen = (uint32_t)rec[1];
  double R[NT][NT], vec[NT], mag = 0.0;
  uint32_t j, k;
  if (rec[0] == 0.0) {
    *pitch = 0;
    for (j = 0; j < ntaps; j++) { coef[j] = 0.0; }
    return 0;
  }
  if (rec[0] != 1.0) { return 4; }
  if (chosen < ntaps / 2 + 1) { return 4; }
  memset(R, 0, sizeof(R));
  for (j = 0; j < ntaps; j++) { for (k = 0; k < ntaps; k++) { R[j][k] = low[(j >= k) ? (j - k) : (k - j)]; } }
  for (j = 0; j < ntaps; j++) { vec[j] = mid[2 + j - ntaps / 2]; }
  if (toeplitz_solve((const double (*)[NT])R, vec, ntaps, 2) != 0) { return 4; }
  for (j = 0; j < ntaps; j++) { mag += fabs(vec[j]); }
  if (mag >= 1.0) {
    for (j = 0; j < ntaps; j++) { vec[j] = 0.0; }
    vec[ntaps / 2] = mid[2] / low[0];
  }
  *pitch = chosen;
  for (j = 0; j < ntaps; j++) { coef[j] = vec[j]; }
  return 0;
}
