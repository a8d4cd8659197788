"""The block certificate's bound, restated in numpy (CPU side of tests/test_gpu_cert.py and tests/tools/cert_study.py).

k_blocks_finish<.., true> (sla_amd/csrc/sla_kernels.hip) accepts the quantised PARCOR codes of a block analysed with
any-order autocorrelation sums when every k_m keeps its distance from the rounding boundaries by
    eps_m = safety * ||a^(m-1)||_1^2 * (1 + |k_m|) / e_(m-1) * (delta + lev_m),
    delta = (n + 64) * 2^-53 * r0,   lev_m = 2 (m + 2) 2^-53 ||a^(m-1)||_1 r0.
"""
import math

import numpy as np

U = 2.0 ** -53
FLT_EPSILON = 1.1920928955078125e-07


def levinson_terms(r, order):
    """float64 Levinson-Durbin that also returns ||a^(m-1)||_1 and e_(m-1) for m = 1..order"""
    a = np.zeros(order + 2)
    a[0] = 1.0
    k = np.zeros(order + 1)
    norm1 = np.zeros(order + 1)
    e_prev = np.zeros(order + 1)
    if abs(r[0]) < FLT_EPSILON:
        return k, norm1, e_prev
    e = r[0]
    for m in range(1, order + 1):
        norm1[m] = np.abs(a[:m]).sum()
        e_prev[m] = e
        num = float(np.dot(a[:m], r[m:0:-1]))
        g = -num / e
        k[m] = -g
        an = a.copy()
        for i in range(1, m):
            an[i] = a[i] + g * a[m - i]
        an[m] = g
        a = an
        e = (1.0 - g * g) * e
        if not e > 0:
            break
    return k, norm1, e_prev


def eps_bound(r, n, order, safety=16.0):
    """eps_m for m = 1..order (index 0 unused); inf where the recursion left the positive-definite range"""
    k, norm1, e_prev = levinson_terms(r, order)
    eps = np.full(order + 1, np.inf)
    delta = (n + 64) * U * r[0]
    for m in range(1, order + 1):
        if not e_prev[m] > 0:
            break
        lev = 2.0 * (m + 2) * U * norm1[m] * r[0]
        eps[m] = safety * norm1[m] ** 2 * (1 + abs(k[m])) / e_prev[m] * (delta + lev)
    return k, eps


def codes(k, order):
    """reference quantiser: src/SLAEncoder.c:573-584 (16 bits for the first three coefficients, 8 for the rest)"""
    out = np.zeros(order + 1, np.int64)
    for m in range(1, order + 1):
        lim = 1 << ((16 if m < 4 else 8) - 1)
        v = k[m] * lim
        rk = math.floor(v + 0.5) if v >= 0 else -math.floor(-v + 0.5)
        out[m] = min(max(rk, -lim), lim - 1)
    return out


def margin(k, m):
    """distance of k_m to the nearest value at which its code changes"""
    lim = 1 << ((16 if m < 4 else 8) - 1)
    v = k * lim
    if v >= lim - 1.5:
        return (v - (lim - 1.5)) / lim
    if v <= -lim + 0.5:
        return ((-lim + 0.5) - v) / lim
    f = abs(v) + 0.5
    d = f - math.floor(f)
    return min(d, 1.0 - d) / lim
