"""The PARCOR lattice kernel's stage forms (sla_amd/csrc/sla_kernels.hip: lattice_chunk_wave_lds) against the oracle's lattice
(reference src/SLAPredictor.c:557-607), through the C-ABI launcher.  Round 4 lets every stage of every wave take the shortest
instruction form its operand bound proves equal to the reference's wrapping 32-bit arithmetic:
    H  high dword of v_mad_i64_i32(k << 17, v, 2^31)          |k| < 2^14 and |k| bnd + 2^14 < 2^31
    S  v_mad_i32_i24(2k, v, 2^15) + SDWA high-word subtract   bnd < 2^23, |k| < 2^22 and |k| bnd + 2^14 < 2^30
    M  v_mad_i32_i24(k, v, 2^14), >> 15, subtract             bnd < 2^23, |k| < 2^22
    W  v_mul_lo_u32, + 2^14, >> 15, subtract                  anything (the reference as it stands)
The cases below are built so that every form runs (a mirror of the kernel's selection rule says which, and the test fails if a
form goes untested), at the edges of each condition, with forms changing from stage to stage inside one wave, on products that
do wrap, on ragged block ends and through the mid/side + shift staging; each case also runs with option lattice_plain (every
stage in form W) and both must equal the oracle bit for bit."""
import ctypes as C
import zlib

import numpy as np
import pytest

import slalibs as S

pytestmark = pytest.mark.gpu

LAT_T = 17          # samples per lane of the lattice wave (kernels/lattice_wave.inc)


@pytest.fixture(scope="module")
def hip():
    import torch
    torch.cuda.init()
    import sla_amd
    sla_amd.lib()
    return sla_amd


@pytest.fixture(scope="module")
def oracle():
    return S.oracle()


class Chunk(C.Structure):
    _fields_ = [("blk_off", C.c_uint64), ("blk_len", C.c_uint32), ("chunk_start", C.c_uint32),
                ("count", C.c_uint32), ("channel", C.c_uint32), ("slot", C.c_uint32), ("int_shift", C.c_uint32)]


class Tuning(C.Structure):
    _fields_ = [("lpc_pack", C.c_uint32), ("lpc_threads", C.c_uint32), ("lpc_blocks_chains", C.c_uint32), ("tail_waves", C.c_uint32),
                ("lpc_tile", C.c_uint32), ("tail_taps", C.c_uint32), ("plan_margin", C.c_double), ("rice_lanes", C.c_uint32),
                ("lattice_plain", C.c_uint32), ("cert_audit", C.c_uint32)]


def forms_of(y, kint, order, per):
    """mirror of lat_pick_form: the set of forms the waves of one block take (y = the lattice input of the whole block)"""
    halo = (order + LAT_T - 1) // LAT_T
    seen = set()
    n = len(y)
    for start in range(0, n, per):
        lo = max(start - halo * LAT_T, 0)
        hi = min(start - halo * LAT_T + 64 * LAT_T, n)
        bnd = int(np.max(np.abs(y[lo:hi].astype(np.int64)))) if hi > lo else 0
        for m in range(1, order + 1):
            ak = abs(int(kint[m]))
            t = ak * bnd + 16384
            t31, t30 = t < 2 ** 31, t < 2 ** 30
            high = t31 and ak < 2 ** 14
            v24 = bnd < 2 ** 23 and ak < 2 ** 22
            seen.add("H" if high else "W" if not v24 else "S" if t30 else "M")
            bnd = min(bnd + ((t >> 15) + 1 if t31 else 65537), 2 ** 31)
    return seen


def run_lattice(hip, x, kint, order, raw, plain, blk_len=None, shift=0):
    import torch
    L = hip.lib()
    n = len(x) if blk_len is None else blk_len
    L.sla_hip_lattice_chunk_samples.restype = C.c_uint32
    per = L.sla_hip_lattice_chunk_samples(order)
    chunks = (Chunk * ((n + per - 1) // per))()
    for i in range(len(chunks)):
        chunks[i] = Chunk(0, n, i * per, min(per, n - i * per), 0, 0, shift)
    d_pcm = torch.from_numpy(np.ascontiguousarray(x)).cuda()
    d_k = torch.from_numpy(np.ascontiguousarray(kint, np.int32)).cuda()
    d_chunks = torch.frombuffer(bytearray(bytes(chunks)), dtype=torch.uint8).cuda()
    d_res = torch.full((len(x),), 0x5A5A5A5A, dtype=torch.int32, device="cuda")
    torch.cuda.synchronize()
    t = Tuning()
    t.lattice_plain = 1 if plain else 0
    L.sla_hip_use_tuning(C.byref(t))
    try:
        fn = L.sla_hip_launch_lattice_raw if raw else L.sla_hip_launch_lattice
        if raw:
            rc = fn(C.c_void_p(d_pcm.data_ptr()), C.c_uint64(len(x)), order, C.c_void_p(d_chunks.data_ptr()), len(chunks),
                    C.c_void_p(d_k.data_ptr()), C.c_void_p(d_res.data_ptr()), None)
        else:
            rc = fn(C.c_void_p(d_pcm.data_ptr()), C.c_uint64(len(x)), 0, order, C.c_void_p(d_chunks.data_ptr()), len(chunks),
                    C.c_void_p(d_k.data_ptr()), C.c_void_p(d_res.data_ptr()), None)
    finally:
        L.sla_hip_use_tuning(None)
    assert rc == 0
    torch.cuda.synchronize()
    return d_res.cpu().numpy(), per


# (name, amplitude bits of the input, |k| range, order, samples): the input is uniform noise of that amplitude
CASES = [
    ("24-bit material, 8-bit coefficients after rshift 8 (C3 / C5)", 23, (1, 127), 32, 5000),
    ("25-bit side channel, rshift 9", 24, (1, 63), 48, 9000),
    ("16-bit material, small coefficients", 14, (1, 16383), 16, 4096),
    ("16-bit material, |k| >= 2^14: high word of the 24-bit multiply-add", 13, (16384, 32767), 16, 4096),
    ("loud 16-bit material, |k| >= 2^14: the bound leaves 2^30 on the way", 15, (16384, 32767), 32, 4096),
    ("full-scale 16-bit, |k| near 2^15: products reach 2^31", 16, (30000, 32767), 8, 3000),
    ("18-bit values with large coefficients: wraps under the 24-bit multiply", 18, (20000, 32767), 24, 4000),
    ("full range: everything wraps", 31, (1, 32767), 32, 5000),
    ("wide values, coefficients at the 2^14 edge", 22, (16380, 16388), 12, 2500),
    ("bound crossing 2^23 between stages", 22, (2000, 16383), 40, 4000),
    ("tiny block", 10, (1, 32767), 4, 7),
    ("order 1", 20, (100, 200), 1, 1500),
    ("order 255 (sixteen halo lanes)", 12, (1, 3000), 255, 3000),
    ("zeros", 0, (1, 32767), 16, 2000),
]


@pytest.mark.parametrize("case", CASES, ids=[c[0] for c in CASES])
def test_lattice_forms_equal_the_reference(oracle, hip, case):
    name, bits, (klo, khi), order, n = case
    rng = np.random.default_rng(zlib.crc32(name.encode()))
    amp = (1 << bits) - 1 if bits else 0
    x = rng.integers(-amp - (1 if bits == 31 else 0), amp + 1, n, dtype=np.int64).astype(np.int32)
    kint = (rng.integers(klo, khi + 1, order + 1) * rng.choice([-1, 1], order + 1)).astype(np.int32)
    kint[0] = 0
    for raw in (True, False):
        y = x if raw else oracle.preemph_i32(x)
        want = oracle.lattice_predict(y, kint)
        for plain in (False, True):
            got, _ = run_lattice(hip, x, kint, order, raw, plain)
            assert np.array_equal(got, want), (name, "raw" if raw else "pre-emphasised", "plain" if plain else "certified forms",
                                               int(np.argmax(got != want)))


def test_every_form_is_exercised(oracle, hip):
    """the cases above reach all four forms (mirror of the kernel's rule), so none of them is green by not running"""
    L = hip.lib()
    L.sla_hip_lattice_chunk_samples.restype = C.c_uint32
    seen = set()
    for name, bits, (klo, khi), order, n in CASES:
        rng = np.random.default_rng(zlib.crc32(name.encode()))
        amp = (1 << bits) - 1 if bits else 0
        x = rng.integers(-amp - (1 if bits == 31 else 0), amp + 1, n, dtype=np.int64).astype(np.int32)
        kint = (rng.integers(klo, khi + 1, order + 1) * rng.choice([-1, 1], order + 1)).astype(np.int32)
        kint[0] = 0
        seen |= forms_of(x, kint, order, L.sla_hip_lattice_chunk_samples(order))
    assert seen == {"H", "S", "M", "W"}, seen


def test_lattice_forms_at_the_edges(oracle, hip):
    """constant-magnitude inputs that put |k| bnd + 2^14 right at 2^30 and 2^31, bnd right at 2^23, and |k| right at 2^14 / 2^22"""
    order, n = 6, 2048
    rng = np.random.default_rng(99)
    sign = rng.choice([-1, 1], n).astype(np.int64)
    for mag, k in [(32767, 32767), (32768, 32767), (32769, 32760), (65535, 16383), (65536, 16384), (131071, 16383), (131072, 16383),
                   ((1 << 23) - 1, 127), (1 << 23, 127), ((1 << 23) - 1, 255), ((1 << 23) + 1, 255), ((1 << 17) - 9, 16383),
                   (255, (1 << 22) - 1), (255, 1 << 22), (3, (1 << 29) + 12345), ((1 << 31) - 1, 1), ((1 << 31) - 1, 16383)]:
        x = (sign * mag).astype(np.int32)
        x[::7] = -x[::7]
        kint = np.array([0] + [k if (i & 1) else -k for i in range(order)], np.int32)
        want = oracle.lattice_predict(x, kint)
        for plain in (False, True):
            got, _ = run_lattice(hip, x, kint, order, True, plain)
            assert np.array_equal(got, want), (mag, k, plain)


def test_lattice_ragged_blocks_and_shift(oracle, hip):
    """block lengths around the wave's (64 - halo lanes) * 17-sample chunks, left-justified input with a shift (the pipeline's staging)"""
    rng = np.random.default_rng(5)
    for order in (8, 16, 32, 48):
        per = (64 - (order + LAT_T - 1) // LAT_T) * LAT_T
        kint = np.concatenate([[0], rng.integers(-120, 121, order)]).astype(np.int32)
        for n in (1, 15, 16, 17, 18, 33, 34, 35, per - 1, per, per + 1, 2 * per, 2 * per + 5, 2048, 4096, 4097, 8192):
            x24 = rng.integers(-(1 << 23), 1 << 23, n, dtype=np.int64).astype(np.int32)
            x = (x24.astype(np.int64) << 8).astype(np.int32)                  # left-justified 24-bit, as the API hands it over
            want = oracle.lattice_predict(oracle.preemph_i32(x24), kint)
            for plain in (False, True):
                got, _ = run_lattice(hip, x, kint, order, False, plain, shift=8)
                assert np.array_equal(got, want), (order, n, plain)
