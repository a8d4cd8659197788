"""GPU tests of the block stage's certified route (sla_hip_launch_lpc_blocks_cert, option "block_cert"): chosen
blocks are analysed with any-order autocorrelation sums, the quantised PARCOR codes (reference
src/SLAEncoder.c:567-589) and the RAW decision (:553-565) are certified per block, the rest is redone by the exact
chain kernels.  What must hold: the bytes, codes, lattice coefficients and residuals are the oracle's on every route;
the doubles of certified blocks sit well inside the certificate's bound; ill-conditioned material falls back."""
import numpy as np
import pytest

import certlib
import slalibs as S
import waveforms as W

pytestmark = pytest.mark.gpu


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


def encode(hip, p, pcm, **options):
    enc = hip.Encoder(p.cap_channels, p.cap_block_samples, p.cap_parcor_order, p.cap_longterm_order, p.cap_lms_order)
    try:
        enc.set_wave_format(p.num_channels, p.bits_per_sample, p.sampling_rate)
        enc.set_encode_parameter(p.parcor_order, p.longterm_order, p.lms_order, p.ch_process_method,
                                 p.window_type, p.max_block_samples)
        enc.set_option("stream", 0)
        for k, v in options.items():
            enc.set_option(k, v)
        data = enc.encode_whole(pcm)
        return data, enc.trace(), enc.last_block_cert()
    finally:
        enc.close()


def tones(n, bits, floor_db, seed, nch=1, rate=48000.0):
    rng = np.random.default_rng(seed)
    t = np.arange(n, dtype=np.float64)
    out = np.zeros((nch, n), np.int32)
    for ch in range(nch):
        x = 0.5 * np.sin(2 * np.pi * (997.0 + 31 * ch) * t / rate) + 0.3 * np.sin(2 * np.pi * 61.0 * t / rate + ch)
        x += (10.0 ** (floor_db / 20.0)) * rng.standard_normal(n)
        full = float(1 << (bits - 1))
        q = np.clip(np.rint(x * full), -full, full - 1).astype(np.int64)
        out[ch] = ((q << (32 - bits)).astype(np.int64)).astype(np.int32)
    return out


def material(name, nch, n, bits, seed):
    if name == "bench":
        return S.synth_pcm(nch, n, bits, seed=12345 + seed)
    if name == "music":
        return W.music_like(nch, n, bits, seed=seed + 1)
    if name.startswith("tones"):
        return tones(n, bits, float(name[5:]), seed, nch)
    return W.gen(name, nch, n, bits, seed=seed)


def check_against_oracle(oracle, p, pcm, data, tr):
    ret, want, to = oracle.encode_trace(p, pcm)
    assert ret == 0
    nb = to.num_blocks
    assert tr.num_blocks == nb
    for f in ("blk_start", "blk_nsmpl", "blk_type", "blk_bytes"):
        assert np.array_equal(getattr(tr, f)[:nb], getattr(to, f)[:nb]), f
    comp = to.blk_type[:nb] == 0
    for f in ("code", "kint", "rshift", "pitch", "rice_init"):
        assert np.array_equal(getattr(tr, f)[:nb][comp], getattr(to, f)[:nb][comp]), f
    assert S.parcor_same(tr, to, nb, comp)
    assert data == want
    return to, comp


CASES = [
    # name, channels, bits, order, max block, ms
    ("bench", 1, 16, 16, 4096, 0), ("bench", 2, 24, 32, 4096, 1), ("bench", 3, 24, 48, 8192, 0),
    ("music", 2, 16, 16, 4096, 1), ("music", 1, 24, 32, 8192, 0), ("music", 2, 24, 48, 16384, 0),
    ("white", 1, 16, 8, 4096, 0), ("gauss", 2, 24, 20, 4096, 1), ("chirp", 1, 24, 32, 8192, 0),
    ("sine", 1, 16, 16, 4096, 0), ("nyquist", 1, 16, 16, 4096, 0), ("posconst", 2, 24, 12, 4096, 0),
    ("tones-20", 1, 24, 32, 4096, 0), ("tones-60", 2, 24, 48, 8192, 0), ("tones-100", 1, 24, 32, 4096, 0),
    ("tones-140", 1, 32, 48, 8192, 0), ("tones-60", 1, 32, 16, 2048, 0), ("silence", 2, 16, 16, 4096, 0),
    ("bench", 1, 8, 4, 2048, 0), ("music", 1, 16, 51, 8192, 0),
]


@pytest.mark.parametrize("name,nch,bits,order,maxb,ms", CASES)
def test_certified_route_is_the_oracle(hip, oracle, name, nch, bits, order, maxb, ms):
    """default route (certificate + exact fallback), forced-exact route and everything-flagged route: same bytes,
    codes and residuals as the oracle; the doubles are the oracle's bit for bit wherever the exact kernels ran"""
    n = 5 * maxb + 1234
    pcm = material(name, nch, n, bits, seed=order + maxb // 1024)
    p = S.make_params(nch, bits, 48000, order, 3 if order > 16 else 1, 8, ms, 1, maxb, cap=(nch, 16384, 52, 3, 8))
    data, tr, (on, redone) = encode(hip, p, pcm)
    assert on == 1
    to, comp = check_against_oracle(oracle, p, pcm, data, tr)
    nslots = int((to.blk_type[:to.num_blocks] != 1).sum()) * nch      # every non-silent (block, channel) is analysed, RAW ones too
    assert int(tr.parcor_exact[:to.num_blocks][comp].sum()) <= redone <= nslots
    # every block through the exact kernels: option off, and certificate on but so strict that nothing certifies
    d2, t2, (on2, _) = encode(hip, p, pcm, block_cert=0)
    assert on2 == 0 and d2 == data and bool(t2.parcor_exact[:to.num_blocks][comp].all())
    assert np.array_equal(t2.parcor[:to.num_blocks].view(np.uint64)[comp], to.parcor[:to.num_blocks].view(np.uint64)[comp])
    d3, t3, (on3, redone3) = encode(hip, p, pcm, block_cert_safety=1e30)
    assert on3 == 1 and d3 == data
    if name != "silence":
        assert redone3 == nslots and bool(t3.parcor_exact[:to.num_blocks][comp].all())
        assert np.array_equal(t3.parcor[:to.num_blocks].view(np.uint64)[comp], to.parcor[:to.num_blocks].view(np.uint64)[comp])


def test_ill_conditioned_material_falls_back(hip, oracle):
    """two pure tones over a -140 dB floor: the Toeplitz system is singular to working precision, no code of the high
    stages can be certified -- the blocks take the exact kernels and the bytes stay the oracle's"""
    pcm = tones(6 * 8192, 24, -140.0, 3)
    p = S.make_params(1, 24, 48000, 48, 3, 8, 0, 1, 8192, cap=(1, 8192, 48, 3, 8))
    data, tr, (on, redone) = encode(hip, p, pcm)
    to, comp = check_against_oracle(oracle, p, pcm, data, tr)
    assert on == 1 and redone >= int(comp.sum()) // 2
    benign = S.synth_pcm(1, 6 * 8192, 24)
    data, tr, (on, redone) = encode(hip, p, benign)
    check_against_oracle(oracle, p, benign, data, tr)
    assert on == 1 and redone == 0


@pytest.mark.parametrize("order,maxb,bits", [(16, 4096, 16), (32, 4096, 24), (48, 8192, 24), (32, 2048, 32)])
def test_certified_doubles_sit_inside_the_bound(hip, oracle, order, maxb, bits):
    """conditioning sweep (applied to the block certificate): tones over noise floors from -20 to
    -120 dB, bench and music-like material.  For every certified (block, channel): |k - k_ref| <= 25 % of the
    certificate's eps_m (safety 16), recomputed here from the oracle's autocorrelation of the same windowed block."""
    win_cache = {}
    worst = 0.0
    certified = flagged = 0
    for si, (name, floor) in enumerate([("bench", 0), ("music", 0)] + [("tones", f) for f in (-20, -40, -60, -80, -100, -120)]):
        n = 12 * maxb
        pcm = tones(n, bits, float(floor), 100 + si) if name == "tones" else material(name, 1, n, bits, seed=si)
        p = S.make_params(1, bits, 48000, order, 1, 8, 0, 1, maxb, cap=(1, maxb, order, 1, 8))
        data, tr, (on, redone) = encode(hip, p, pcm)
        to, comp = check_against_oracle(oracle, p, pcm, data, tr)
        for b in np.nonzero(comp)[0]:
            if tr.parcor_exact[b, 0]:
                flagged += 1
                continue
            s, nb = int(to.blk_start[b]), int(to.blk_nsmpl[b])
            if nb not in win_cache:
                win_cache[nb] = oracle.window(1, nb)
            x = oracle.preemph_f64(pcm[0, s:s + nb].astype(np.float64) * 2.0 ** -31 * win_cache[nb])
            r = oracle.autocorr(x, order + 1)
            if r[0] < certlib.FLT_EPSILON:
                continue
            _, eps = certlib.eps_bound(r, nb, order, safety=16.0)
            dk = np.abs(tr.parcor[b, 0, 1:] - to.parcor[b, 0, 1:])
            assert np.all(np.isfinite(eps[1:]))
            worst = max(worst, float(np.max(dk / eps[1:])))
            certified += 1
    assert certified >= 40 and worst <= 0.25, (certified, flagged, worst)


# ------------------------------------------------------------------ the partition search's certificate

@pytest.mark.parametrize("order,bits", [(16, 24), (32, 24), (48, 24), (32, 32)])
def test_search_certificate_conditioning_sweep(hip, oracle, order, bits):
    """>= 10^4 windows in all -- pure tones over noise floors from -20 to -140 dB, 24- and 32-bit
    material, orders 16 / 32 / 48, every window forced over the exactness limit.  For every candidate that carries a
    finite bracket the reference's value (autocorrelation in ITS order, ITS Levinson-Durbin recursion) must sit within 25 %
    of the bracket's half width."""
    import ctypes as C
    import torch
    L = hip.lib()
    L.sla_hip_search_exact_lags.restype = C.c_uint32
    lags = L.sla_hip_search_exact_lags(order)
    W_ = 4096
    cand = sorted({(i * 1024, (j - i) * 1024) for i in range(5) for j in range(i + 2, 5)})
    nwin = 420

    class Group(C.Structure):
        _fields_ = [("pcm_off", C.c_uint64)] + [(k, C.c_uint32) for k in (
            "num_samples", "channel", "win_off", "int_shift", "cand_first", "cand_count", "slot_first", "pad_")]

    worst, bracketed, total = 0.0, 0, 0
    for fi, floor in enumerate((-20, -40, -60, -90, -120, -140)):
        n = W_ * nwin
        pcm = tones(n, bits, float(floor), 1000 * order + fi)
        groups = (Group * nwin)(*[Group(g * W_, W_, 0, 0xFFFFFFFF, 32 - bits, 0, len(cand), g * len(cand), 0) for g in range(nwin)])
        d_pcm = torch.from_numpy(pcm).cuda()
        d_g = torch.frombuffer(bytearray(bytes(groups)), dtype=torch.uint8).cuda()
        d_c = torch.from_numpy(np.array(cand, np.uint32)).cuda()
        d_ts = torch.zeros(nwin * 16 * 2 * lags, dtype=torch.float64, device="cuda")
        d_out = torch.zeros(nwin * len(cand) * (order + 2), dtype=torch.float64, device="cuda")
        torch.cuda.synchronize()
        rc = L.sla_hip_launch_search_exact(C.c_void_p(d_pcm.data_ptr()), C.c_uint64(n), 0, order, C.c_void_p(d_g.data_ptr()), nwin, W_,
                                           len(cand), C.c_void_p(d_c.data_ptr()), C.c_void_p(d_ts.data_ptr()), C.c_void_p(d_out.data_ptr()),
                                           C.c_double(0.0), C.c_double(64.0), None, None)
        assert rc == 0
        torch.cuda.synchronize()
        out = d_out.cpu().numpy().reshape(nwin, len(cand), order + 2)
        xd = pcm[0].astype(np.float64) * 2.0 ** -31
        for g in range(nwin):
            for i, (s, ln) in enumerate(cand):
                o = out[g, i]
                total += 1
                if not np.isfinite(o[1]):
                    continue                                       # no bracket: the window is rerun as serial chains
                xs = np.ascontiguousarray(xd[g * W_ + s:g * W_ + s + ln])
                r0 = oracle.autocorr(xs, 1)[0]
                _, par = oracle.parcor(xs, order)
                ref = np.log2(r0) + np.sum(np.log2(1.0 - par[1:] ** 2))
                mid = np.log2(o[0]) + o[2]
                assert abs(ref - mid) <= 0.25 * o[1], (order, bits, floor, g, s, ln, ref - mid, o[1])
                worst = max(worst, abs(ref - mid) / o[1])
                bracketed += 1
    assert total == 6 * nwin * len(cand) and bracketed >= total // 3, (total, bracketed, worst)


# ------------------------------------------------------------------ the standing audit of the certificate (round 4)

AUDIT_CASES = [("bench", 2, 24, 32, 4096, 1), ("music", 2, 16, 16, 4096, 1), ("bench", 3, 24, 48, 8192, 0), ("tones-60", 2, 24, 48, 8192, 0),
               ("tones-100", 1, 24, 32, 4096, 0), ("white", 1, 16, 8, 4096, 0)]


@pytest.mark.parametrize("name,nch,bits,order,maxb,ms", AUDIT_CASES)
def test_audited_certificates_hold(hip, oracle, name, nch, bits, order, maxb, ms):
    """option cert_audit = 1: EVERY certified (block, channel) is analysed by the exact chain kernels as well, which compare
    codes, lattice coefficients and the RAW side with what the certified run stored -- no difference, the bytes stay the
    oracle's, the count of pairs the certificate itself handed over does not change; cert_audit = 3: every third pair"""
    n = 6 * maxb + 777
    pcm = material(name, nch, n, bits, seed=order + 3)
    p = S.make_params(nch, bits, 48000, order, 3 if order > 16 else 1, 8, ms, 1, maxb, cap=(nch, 16384, 52, 3, 8))
    data, tr, (on, redone) = encode(hip, p, pcm)
    to, comp = check_against_oracle(oracle, p, pcm, data, tr)
    nb = to.num_blocks
    certified = int((~tr.parcor_exact[:nb][comp].astype(bool)).sum())
    for every in (1, 3):
        enc = hip.Encoder(p.cap_channels, p.cap_block_samples, p.cap_parcor_order, p.cap_longterm_order, p.cap_lms_order)
        try:
            enc.set_wave_format(p.num_channels, p.bits_per_sample, p.sampling_rate)
            enc.set_encode_parameter(p.parcor_order, p.longterm_order, p.lms_order, p.ch_process_method, p.window_type, p.max_block_samples)
            enc.set_option("stream", 0)
            enc.set_option("cert_audit", every)
            d2 = enc.encode_whole(pcm)
            t2 = enc.trace()
            ok, bad = enc.last_cert_audit()
            on2, redone2 = enc.last_block_cert()
        finally:
            enc.close()
        assert d2 == data and bad == 0 and on2 == 1 and redone2 == redone
        check_against_oracle(oracle, p, pcm, d2, t2)
        if every == 1:
            # every certified pair of a compressed block was audited (RAW blocks' pairs are analysed and audited too: >=), and
            # the doubles the trace shows are now the exact kernels' everywhere
            assert ok >= certified and bool(t2.parcor_exact[:nb][comp].all())
            assert np.array_equal(t2.parcor[:nb].view(np.uint64)[comp], to.parcor[:nb].view(np.uint64)[comp])
        else:
            assert 0 < ok or certified < 3


def _one_block_params(bits, order, n):
    return S.make_params(1, bits, 48000, order, 1, 8, 0, 1, n, cap=(1, 16384, 52, 3, 8))


def boundary_cases(oracle, order, bits, n=4096):
    """[(pcm, coefficient index m, distance of k_m * 2^(q-1) from the half-integer it was steered to)], all by the oracle's doubles
    (= the reference's).  Two steps: a mixing gain between two signals is bisected until the two neighbouring integer signals
    straddle the boundary (that leaves ~1e-7 of a code step: one sample moving by one quantisation step shifts k_m by about
    that much); then single samples are nudged by one step each -- every one moves k_m by its own small amount, measured one
    by one, and a random search over subsets of 48 of them picks a combination that lands within ~1e-11 (twice)."""
    rng = np.random.default_rng(order * 100 + bits)
    base = W.music_like(1, n, bits, seed=order)[0].astype(np.int64) >> (32 - bits)
    other = np.rint(rng.standard_normal(n) * (1 << (bits - 6))).astype(np.int64)
    full = 1 << (bits - 1)
    p = _one_block_params(bits, order, n)

    def quantised(g):
        return np.clip(np.rint(base + g * other), -full, full - 1).astype(np.int64)

    def pcm_of(q):
        return ((q << (32 - bits)).astype(np.int64)).astype(np.int32)[None, :]

    def value(q, m, lim):
        ret, want, to = oracle.encode_trace(p, pcm_of(q))
        assert ret == 0
        if to.num_blocks != 1 or to.blk_type[0] != 0:
            return None
        return to.parcor[0][0][m] * lim

    out = []
    for m in sorted({2, 5, order // 2, order}):
        lim = float(1 << ((16 if m < 4 else 8) - 1))
        v0 = value(quantised(0.0), m, lim)
        if v0 is None:
            continue
        target = np.floor(v0) + 0.5
        lo, hi, flo = 0.0, None, v0 - target
        for g in np.geomspace(1e-4, 2.0, 40):
            v1 = value(quantised(g), m, lim)
            if v1 is None:
                break
            if (v1 - target > 0) != (flo > 0):
                hi = g
                break
            lo, flo = g, v1 - target
        if hi is None:
            continue
        for _ in range(200):
            mid = 0.5 * (lo + hi)
            if np.array_equal(quantised(mid), quantised(lo)) or np.array_equal(quantised(mid), quantised(hi)):
                break
            vm = value(quantised(mid), m, lim)
            if vm is None:
                break
            if (vm - target > 0) == (flo > 0):
                lo, flo = mid, vm - target
            else:
                hi = mid
        q = quantised(lo)
        f = value(q, m, lim) - target
        for _ in range(2):
            idx = rng.choice(np.nonzero(np.abs(q) < full - 2)[0], 48, replace=False)
            step = rng.choice([-1, 1], 48)
            d = np.zeros(48)
            for j in range(48):
                q2 = q.copy(); q2[idx[j]] += step[j]
                v2 = value(q2, m, lim)
                d[j] = (v2 - target - f) if v2 is not None else 1e9
            pick = rng.integers(0, 2, (200000, 48)).astype(np.float64)
            best = pick[np.argmin(np.abs(f + pick @ d))] > 0
            q3 = q.copy(); q3[idx[best]] += step[best]
            v3 = value(q3, m, lim)
            if v3 is not None and abs(v3 - target) < abs(f):
                q, f = q3, v3 - target
        out.append((pcm_of(q), m, abs(f)))
        # and its neighbour on the other side of the boundary: one more sample, the one that crosses by the least
        idx = rng.choice(np.nonzero(np.abs(q) < full - 2)[0], 24, replace=False)
        cross = None
        for j in idx:
            for st in (-1, 1):
                q2 = q.copy(); q2[j] += st
                v2 = value(q2, m, lim)
                if v2 is not None and ((v2 - target) > 0) != (f > 0) and (cross is None or abs(v2 - target) < cross[1]):
                    cross = (q2, abs(v2 - target))
        if cross is not None:
            out.append((pcm_of(cross[0]), m, cross[1]))
    return p, out


@pytest.mark.parametrize("order,bits", [(16, 24), (32, 24), (48, 24), (16, 16)])
def test_codes_next_to_a_rounding_boundary(hip, oracle, order, bits):
    """Adversarial input for the certificate: blocks whose k_m * 2^(q-1) -- the oracle's doubles,
    i.e. the reference's -- sits as close to a half-integer as integer PCM allows (boundary_cases).  On both sides of the
    boundary the HIP path must give the oracle's bytes and codes; a pair closer than the narrowest width the certificate can
    have (>= 16 (n + 64) 2^-53 2^(q-1) ~ 1e-9 of a code step) must have been handed to the exact kernels; and with every
    certificate audited nothing differs."""
    p, cases = boundary_cases(oracle, order, bits)
    assert cases, "no boundary case was constructed"
    best = min(d for _, _, d in cases)
    assert best < (1e-11 if bits >= 24 else 1e-7), best
    for pcm, m, dist in cases:
        ret, want, to = oracle.encode_trace(p, pcm)
        assert ret == 0
        data, tr, (on, redone) = encode(hip, p, pcm)
        assert on == 1 and data == want, (order, bits, m, dist)
        assert np.array_equal(tr.code[:1], to.code[:1]) and np.array_equal(tr.kint[:1], to.kint[:1])
        if dist < 2e-10:
            assert bool(tr.parcor_exact[0][0]) and redone >= 1, (order, bits, m, dist)
        d2, t2, _ = encode(hip, p, pcm, cert_audit=1)
        assert d2 == want
