"""GPU tests of the per-call predictor / coder API (include/SLAPredictor.h, include/SLACoder.h): each entry point,
called through the C-ABI of libsla_hip.so exactly as src/SLAEncoder.c calls the reference's, against the CPU oracle
-- bit for bit (doubles by bit pattern)."""
import ctypes as C

import numpy as np
import pytest

import slalibs as S
import waveforms as W

pytestmark = pytest.mark.gpu

f64p, i32p, u32p = C.POINTER(C.c_double), C.POINTER(C.c_int32), C.POINTER(C.c_uint32)


def p(a, t):
    return a.ctypes.data_as(t)


@pytest.fixture(scope="module")
def L():
    import torch
    torch.cuda.init()
    import sla_amd
    lib = sla_amd.lib()
    for name in ("SLALPCCalculator_Create", "SLALPCSynthesizer_Create", "SLALongTermCalculator_Create", "SLALongTermSynthesizer_Create",
                 "SLALMSFilter_Create", "SLAOptimalEncodeEstimator_Create", "SLAEmphasisFilter_Create", "SLACoder_Create"):
        getattr(lib, name).restype = C.c_void_p
    lib.SLAEmphasisFilter_PreEmphasisDouble.restype = None
    lib.SLACoder_CalculateInitialRecursiveRiceParameter.restype = None
    for name in ("SLALPCCalculator_Destroy", "SLALPCSynthesizer_Destroy", "SLALongTermCalculator_Destroy", "SLALongTermSynthesizer_Destroy",
                 "SLALMSFilter_Destroy", "SLAOptimalEncodeEstimator_Destroy", "SLAEmphasisFilter_Destroy", "SLACoder_Destroy"):
        getattr(lib, name).restype = None
        getattr(lib, name).argtypes = [C.c_void_p]
    return lib


def windowed(n, bits, seed, oracle):
    x = W.music_like(1, n, bits, seed=seed)[0].astype(np.float64) * 2.0 ** -31
    w = oracle.window(1, n)
    return oracle.preemph_f64(np.ascontiguousarray(x * w))


@pytest.mark.parametrize("n,order", [(4096, 16), (2048, 8), (16384, 48), (3000, 32), (700, 5), (10, 16), (17, 16), (1, 1)])
def test_parcor_and_code_length(oracle, L, n, order):
    x = windowed(n, 16, n + order, oracle)
    h = L.SLALPCCalculator_Create(48)
    assert h
    par = np.zeros(order + 1)
    rc = L.SLALPCCalculator_CalculatePARCORCoefDouble(C.c_void_p(h), p(x, f64p), n, p(par, f64p), order)
    ret, want = oracle.parcor(x, order)
    assert rc == ret == 0
    assert np.array_equal(par.view(np.uint64), want.view(np.uint64))
    out = C.c_double()
    rc = L.SLALPCCalculator_EstimateCodeLength(p(x, f64p), n, 16, p(par, f64p), order, C.byref(out))
    assert rc == 0 and out.value.hex() == float(oracle.code_length(x, 16, want)).hex()
    # argument checks as the reference: NULL, order beyond the handle
    assert L.SLALPCCalculator_CalculatePARCORCoefDouble(C.c_void_p(h), None, n, p(par, f64p), order) == 2
    assert L.SLALPCCalculator_CalculatePARCORCoefDouble(C.c_void_p(h), p(x, f64p), n, p(par, f64p), 49) == 3
    L.SLALPCCalculator_Destroy(C.c_void_p(h))


@pytest.mark.parametrize("n,order", [(4096, 16), (5000, 32), (9000, 48), (100, 8), (5, 16), (1008, 1)])
def test_lattice_predict(oracle, L, n, order):
    rng = np.random.default_rng(n)
    x = oracle.preemph_i32(W.music_like(1, n, 24, seed=order)[0] >> 8)
    kint = rng.integers(-32768, 32767, order + 1, dtype=np.int64).astype(np.int32)
    kint[0] = 0
    h = L.SLALPCSynthesizer_Create(48)
    res = np.zeros(n, np.int32)
    assert L.SLALPCSynthesizer_PredictByParcorCoefInt32(C.c_void_p(h), p(x, i32p), n, p(kint, i32p), order, p(res, i32p)) == 0
    assert np.array_equal(res, oracle.lattice_predict(x, kint))
    # a second block without a reset is refused, after a reset it runs again
    assert L.SLALPCSynthesizer_PredictByParcorCoefInt32(C.c_void_p(h), p(x, i32p), n, p(kint, i32p), order, p(res, i32p)) == 1
    assert L.SLALPCSynthesizer_Reset(C.c_void_p(h)) == 0
    res2 = np.zeros(n, np.int32)
    assert L.SLALPCSynthesizer_PredictByParcorCoefInt32(C.c_void_p(h), p(x, i32p), n, p(kint, i32p), order, p(res2, i32p)) == 0
    assert np.array_equal(res, res2)
    L.SLALPCSynthesizer_Destroy(C.c_void_p(h))


def test_emphasis_filters(oracle, L):
    x = W.music_like(1, 10000, 24, seed=3)[0] >> 8
    want = oracle.preemph_i32(x.copy())
    h = L.SLAEmphasisFilter_Create()
    got = x.copy()
    # two calls: the filter carries its previous sample across them like the reference
    assert L.SLAEmphasisFilter_PreEmphasisInt32(C.c_void_p(h), p(got[:3333], i32p), 3333, 5) == 0
    tail = np.ascontiguousarray(got[3333:])
    assert L.SLAEmphasisFilter_PreEmphasisInt32(C.c_void_p(h), p(tail, i32p), len(tail), 5) == 0
    got[3333:] = tail
    assert np.array_equal(got, want)
    L.SLAEmphasisFilter_Destroy(C.c_void_p(h))
    d = x.astype(np.float64) * 2.0 ** -31
    wantd = oracle.preemph_f64(d.copy())
    L.SLAEmphasisFilter_PreEmphasisDouble(p(d, f64p), len(d), 5)
    assert np.array_equal(d.view(np.uint64), wantd.view(np.uint64))


@pytest.mark.parametrize("n,taps,seed", [(4096, 1, 1), (4096, 3, 2), (8192, 5, 3), (2500, 3, 4), (300, 1, 5)])
def test_long_term_analysis_and_filter(oracle, L, n, taps, seed):
    t = np.arange(n)
    res = (20000 * np.sin(2 * np.pi * t / (37 + seed)) + W.music_like(1, n, 16, seed=seed)[0].astype(np.float64) / 65536 * 0.2).astype(np.int32) << 8
    fft = 16384 if n <= 8192 else 32768
    h = L.SLALongTermCalculator_Create(fft, 256, 10, 5)
    assert h
    pitch = C.c_uint32()
    coef = np.zeros(5)
    rc = L.SLALongTermCalculator_CalculateCoef(C.c_void_p(h), p(res, i32p), n, C.byref(pitch), p(coef, f64p), taps)
    ret, wp, wc = oracle.ltm_analyze(res, fft, taps)
    assert rc == ret
    if ret == 0:
        assert pitch.value == wp and np.array_equal(coef[:taps].view(np.uint64), wc.view(np.uint64))
    L.SLALongTermCalculator_Destroy(C.c_void_p(h))
    if ret == 0 and wp >= 3:
        q = np.array([int(np.floor(c * 32768 + 0.5)) << 16 if c >= 0 else -(int(np.floor(-c * 32768 + 0.5)) << 16) for c in wc] + [0] * 5, np.int64).astype(np.int32)[:5]
        s = L.SLALongTermSynthesizer_Create(5, 256)
        out = np.zeros(n, np.int32)
        assert L.SLALongTermSynthesizer_PredictInt32(C.c_void_p(s), p(res, i32p), n, wp, p(q, i32p), taps, p(out, i32p)) == 0
        assert np.array_equal(out, oracle.ltm_predict(res, wp, q[:taps]))
        L.SLALongTermSynthesizer_Destroy(C.c_void_p(s))


@pytest.mark.parametrize("n,order", [(4096, 8), (5000, 4), (3000, 16), (2000, 32), (6, 8)])
def test_lms_predict(oracle, L, n, order):
    x = W.music_like(1, n, 16, seed=n)[0] >> 18
    h = L.SLALMSFilter_Create(32)
    out = np.zeros(n, np.int32)
    assert L.SLALMSFilter_PredictInt32(C.c_void_p(h), order, p(x, i32p), n, p(out, i32p)) == 0
    assert np.array_equal(out, oracle.lms_predict(x, order))
    assert L.SLALMSFilter_PredictInt32(C.c_void_p(h), order, p(x, i32p), n, p(out, i32p)) == 1      # not reset
    L.SLALMSFilter_Destroy(C.c_void_p(h))


@pytest.mark.parametrize("nch,n,maxb,order", [(1, 4096, 4096, 16), (2, 16384, 16384, 8), (2, 12288, 8192, 32), (1, 5000, 4096, 16), (3, 8192, 8192, 4)])
def test_partition_search(oracle, L, nch, n, maxb, order):
    pcm = W.music_like(nch, n, 16, seed=n + nch)
    pcm[:, n // 2:] >>= 3                       # two regimes: a partition point worth finding
    d = np.ascontiguousarray(pcm.astype(np.float64) * 2.0 ** -31)
    lp = L.SLALPCCalculator_Create(48)
    oe = L.SLAOptimalEncodeEstimator_Create(16384, 1024)
    rows = (f64p * nch)(*[p(d[c], f64p) for c in range(nch)])
    npart = C.c_uint32()
    parts = np.zeros(32, np.uint32)
    rc = L.SLAOptimalEncodeEstimator_SearchOptimalBlockPartitions(C.c_void_p(oe), C.c_void_p(lp), rows, nch, n, 2048, 1024, maxb, 16, order,
                                                                   C.byref(npart), p(parts, u32p))
    ret, want = oracle.partition_search(d, 2048, 1024, maxb, 16, order)
    assert rc == ret == 0
    assert list(parts[:npart.value]) == list(want)
    assert L.SLAOptimalEncodeEstimator_CalculateMaxNumPartitions(16384, 1024) == 17
    L.SLAOptimalEncodeEstimator_Destroy(C.c_void_p(oe))
    L.SLALPCCalculator_Destroy(C.c_void_p(lp))


def test_coder_initial_parameters(oracle, L):
    nch, n = 3, 5000
    res = W.music_like(nch, n, 16, seed=8) >> 20
    res[2] = 0                                  # mean 0 -> parameter 1
    c = L.SLACoder_Create(8, 2)
    rows = (i32p * nch)(*[p(np.ascontiguousarray(res[ch]), i32p) for ch in range(nch)])
    keep = [np.ascontiguousarray(res[ch]) for ch in range(nch)]
    rows = (i32p * nch)(*[p(k, i32p) for k in keep])
    L.SLACoder_CalculateInitialRecursiveRiceParameter(C.c_void_p(c), 2, rows, nch, n)
    want = oracle.rice_init(np.ascontiguousarray(res))
    got = [L.sla_hip_coder_initial_parameter(C.c_void_p(c), ch) for ch in range(nch)]
    assert got == list(want)
    L.SLACoder_Destroy(C.c_void_p(c))


# ------------------------------------------------------------------ decode side: Synthesize* / DeEmphasis

@pytest.mark.parametrize("n,order", [(4096, 16), (5000, 32), (9000, 48), (100, 8), (5, 16), (1008, 1), (3000, 64), (2500, 100), (2048, 255)])
def test_lattice_synthesize_inverts_predict(oracle, L, n, order):
    rng = np.random.default_rng(n + order)
    x = oracle.preemph_i32(W.music_like(1, n, 24, seed=order)[0] >> 8)
    kint = (rng.integers(-32768, 32767, order + 1, dtype=np.int64) >> int(rng.integers(0, 3))).astype(np.int32)
    kint[0] = 0
    res = oracle.lattice_predict(x, kint)
    h = L.SLALPCSynthesizer_Create(255)
    out = np.zeros(n, np.int32)
    assert L.SLALPCSynthesizer_SynthesizeByParcorCoefInt32(C.c_void_p(h), p(res, i32p), n, p(kint, i32p), order, p(out, i32p)) == 0
    assert np.array_equal(out, oracle.lattice_synth(res, kint)) and np.array_equal(out, x)
    assert L.SLALPCSynthesizer_SynthesizeByParcorCoefInt32(C.c_void_p(h), p(res, i32p), n, p(kint, i32p), order, p(out, i32p)) == 1   # no reset
    assert L.SLALPCSynthesizer_Reset(C.c_void_p(h)) == 0
    assert L.SLALPCSynthesizer_SynthesizeByParcorCoefInt32(C.c_void_p(h), None, n, p(kint, i32p), order, p(out, i32p)) == 2
    assert L.SLALPCSynthesizer_SynthesizeByParcorCoefInt32(C.c_void_p(h), p(res, i32p), n, p(kint, i32p), 256, p(out, i32p)) == 3
    L.SLALPCSynthesizer_Destroy(C.c_void_p(h))


@pytest.mark.parametrize("n,pitch,taps", [(4096, 131, 3), (5000, 37, 5), (4096, 3, 1), (8192, 255, 3), (300, 40, 5), (4096, 5, 5), (16384, 100, 1)])
def test_longterm_synthesize_inverts_predict(oracle, L, n, pitch, taps):
    rng = np.random.default_rng(pitch)
    x = W.music_like(1, n, 16, seed=pitch)[0] >> 16
    coef = (rng.integers(-20000, 20000, taps).astype(np.int64) << 16).astype(np.int32)
    res = oracle.ltm_predict(x, pitch, coef)
    h = L.SLALongTermSynthesizer_Create(5, 256)
    out = np.zeros(n, np.int32)
    assert L.SLALongTermSynthesizer_SynthesizeInt32(C.c_void_p(h), p(res, i32p), n, pitch, p(coef, i32p), taps, p(out, i32p)) == 0
    assert np.array_equal(out, oracle.ltm_synth(res, pitch, coef)) and np.array_equal(out, x)
    assert L.SLALongTermSynthesizer_SynthesizeInt32(C.c_void_p(h), p(res, i32p), n, pitch, p(coef, i32p), taps, p(out, i32p)) == 1
    assert L.SLALongTermSynthesizer_Reset(C.c_void_p(h)) == 0
    out0 = np.zeros(n, np.int32)
    assert L.SLALongTermSynthesizer_SynthesizeInt32(C.c_void_p(h), p(res, i32p), n, 0, p(coef, i32p), taps, p(out0, i32p)) == 0   # pitch 0: copy
    assert np.array_equal(out0, res)
    L.SLALongTermSynthesizer_Destroy(C.c_void_p(h))


@pytest.mark.parametrize("n,order", [(4096, 8), (5000, 4), (3000, 16), (2049, 32), (7, 8), (3, 4)])
def test_lms_synthesize_inverts_predict(oracle, L, n, order):
    x = W.music_like(1, n, 16, seed=n)[0] >> 16
    res = oracle.lms_predict(x, order)
    h = L.SLALMSFilter_Create(32)
    out = np.zeros(n, np.int32)
    assert L.SLALMSFilter_SynthesizeInt32(C.c_void_p(h), order, p(res, i32p), n, p(out, i32p)) == 0
    assert np.array_equal(out, oracle.lms_synth(res, order)) and np.array_equal(out, x)
    assert L.SLALMSFilter_SynthesizeInt32(C.c_void_p(h), order, p(res, i32p), n, p(out, i32p)) == 1
    assert L.SLALMSFilter_Reset(C.c_void_p(h)) == 0
    assert L.SLALMSFilter_SynthesizeInt32(C.c_void_p(h), 64, p(res, i32p), n, p(out, i32p)) == 3
    L.SLALMSFilter_Destroy(C.c_void_p(h))


def test_deemphasis_inverts_preemphasis_across_calls(oracle, L):
    x = W.music_like(1, 9000, 16, seed=4)[0] >> 16
    pre = oracle.preemph_i32(x)
    assert np.array_equal(oracle.deemph_i32(pre), x)
    h = L.SLAEmphasisFilter_Create()
    got = pre.copy()
    a, b = got[:4000], got[4000:]                      # two calls: the filter carries its last output over
    assert L.SLAEmphasisFilter_DeEmphasisInt32(C.c_void_p(h), p(a, i32p), 4000, 5) == 0
    assert L.SLAEmphasisFilter_DeEmphasisInt32(C.c_void_p(h), p(b, i32p), 5000, 5) == 0
    assert np.array_equal(got, x)
    assert L.SLAEmphasisFilter_DeEmphasisInt32(C.c_void_p(h), None, 10, 5) == 2
    L.SLAEmphasisFilter_Destroy(C.c_void_p(h))
