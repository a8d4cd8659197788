"""Pins the CPU oracle (oracle/sla_oracle.c) against the UNMODIFIED reference compiled into
oracle/_ref (build container only; skipped where the reference sources never existed).

Every comparison is bit-exact: doubles are compared by bit pattern, bytes by equality.
Mirrors the reference's own suites: test_SLAPredictor.c (autocorrelation :1010-1075, LPC sanity
:75-209, predict/synth identity :437-552, Dijkstra known answers :807-990), test_SLACoder.c
(put/get identity :25-278), test_SLAUtility.c (CRC16 known answers :41-72) and
test_SLAEncodeDecode.c (round-trip matrix :558-1172)."""
import numpy as np
import pytest

import slalibs as S
import waveforms as W

A_WAV = "/root/reference/test/a.wav"


def bits_equal(a, b):
    a = np.ascontiguousarray(a, np.float64)
    b = np.ascontiguousarray(b, np.float64)
    return a.shape == b.shape and np.array_equal(a.view(np.uint64), b.view(np.uint64))


def to_double(pcm_row):
    return pcm_row.astype(np.float64) * 2.0 ** -31


# ---------------------------------------------------------------- unit level

@pytest.mark.parametrize("name", W.NAMES)
@pytest.mark.parametrize("n,lags", [(256, 256), (4096, 33), (3072, 17), (2048, 49), (1000, 9), (37, 17)])
def test_autocorr(oracle, ref, name, n, lags):
    x = to_double(W.gen(name, 1, n, 24, seed=n)[0])
    assert bits_equal(oracle.autocorr(x, lags), ref.autocorr(x, lags))


def test_autocorr_windowed(oracle, ref):
    x = to_double(W.music_like(1, 4096, 24)[0]) * ref.window(1, 4096)
    x = ref.preemph_f64(x)
    assert bits_equal(oracle.autocorr(x, 33), ref.autocorr(x, 33))


@pytest.mark.parametrize("name", W.NAMES)
@pytest.mark.parametrize("order", [1, 4, 8, 16, 32, 48])
def test_parcor(oracle, ref, name, order):
    x = to_double(W.gen(name, 1, 2048, 16, seed=order)[0])
    ro, po = oracle.parcor(x, order)
    rr, pr = ref.parcor(x, order)
    assert ro == rr == 0
    assert bits_equal(po, pr)
    assert oracle.code_length(x, 16, po).hex() == ref.code_length(x, 16, pr).hex()


def test_parcor_short_input(oracle, ref):
    x = to_double(W.gen("white", 1, 10, 16)[0])
    _, po = oracle.parcor(x, 16)
    _, pr = ref.parcor(x, 16)
    assert bits_equal(po, pr) and not po.any()


def test_lpc_sanity(oracle):
    """reference test_SLAPredictor.c:75-209"""
    _, p = oracle.parcor(np.zeros(1024), 8)
    assert not p.any()
    _, lpc, _ = oracle.levinson(oracle.autocorr(np.full(1024, 0.5), 2), 1)
    assert abs(lpc[1] + 1.0) < 0.01
    nyq = np.where(np.arange(1024) % 2 == 0, 1.0, -1.0)
    _, lpc, _ = oracle.levinson(oracle.autocorr(nyq, 2), 1)
    assert abs(lpc[1] - 1.0) < 0.01


@pytest.mark.parametrize("name", W.NAMES)
@pytest.mark.parametrize("bits,order", [(16, 16), (24, 16), (16, 256), (24, 32), (24, 48)])
def test_lattice(oracle, ref, name, bits, order):
    n = 8192 if order < 256 else 2048
    pcm = W.gen(name, 1, n, bits, seed=3)[0] >> (32 - bits)
    x = to_double(W.gen(name, 1, n, bits, seed=3)[0])
    _, par = ref.parcor(x, order)
    rshift = max(bits - 16, 0)
    kint = np.zeros(order + 1, np.int32)
    kint[1:] = (np.rint(np.nan_to_num(par[1:]) * 32767).astype(np.int32)) >> rshift
    a, b = oracle.lattice_predict(pcm, kint), ref.lattice_predict(pcm, kint)
    assert np.array_equal(a, b)
    assert np.array_equal(oracle.lattice_synth(a, kint), ref.lattice_synth(b, kint))
    assert np.array_equal(oracle.lattice_synth(a, kint), pcm)      # predict o synth = id


def test_lattice_wraparound(oracle, ref):
    """products that overflow int32 must wrap identically."""
    rng = np.random.default_rng(7)
    x = rng.integers(-2 ** 31, 2 ** 31 - 1, 4096, dtype=np.int64).astype(np.int32)
    kint = rng.integers(-32768, 32767, 33, dtype=np.int64).astype(np.int32)
    kint[0] = 0
    assert np.array_equal(oracle.lattice_predict(x, kint), ref.lattice_predict(x, kint))


@pytest.mark.parametrize("name", W.NAMES)
def test_emphasis(oracle, ref, name):
    pcm = W.gen(name, 1, 4096, 24, seed=5)[0] >> 8
    a = oracle.preemph_i32(pcm)
    assert np.array_equal(a, ref.preemph_i32(pcm))
    assert np.array_equal(oracle.deemph_i32(a), pcm)
    assert np.array_equal(ref.deemph_i32(a), pcm)
    x = to_double(W.gen(name, 1, 4096, 24, seed=5)[0])
    assert bits_equal(oracle.preemph_f64(x), ref.preemph_f64(x))


@pytest.mark.parametrize("wtype", [0, 1, 2, 3, 4])
@pytest.mark.parametrize("n", [1, 2, 1024, 4096, 3000])
def test_window(oracle, ref, wtype, n):
    assert bits_equal(oracle.window(wtype, n), ref.window(wtype, n))


@pytest.mark.parametrize("n", [8, 64, 4096, 32768])
def test_fft(oracle, ref, n):
    rng = np.random.default_rng(n)
    x = rng.standard_normal(n)
    f_o, f_r = oracle.fft(x, 1), ref.fft(x, 1)
    assert bits_equal(f_o, f_r)
    assert bits_equal(oracle.fft(f_o, -1), ref.fft(f_r, -1))


def test_crc16_known_answers(oracle, ref):
    """reference test_SLAUtility.c:41-46, 71-72"""
    assert oracle.crc16(b"123456789") == 0xBB3D == ref.crc16(b"123456789")
    assert oracle.crc16(b"") == 0
    data = open(A_WAV, "rb").read()
    assert oracle.crc16(data) == 0xA611 == ref.crc16(data)
    png = open("/root/reference/test/PriChanIcon.png", "rb").read()
    assert oracle.crc16(png) == 0xEA63


@pytest.mark.parametrize("dim", [1, 3, 5])
def test_lesolve(oracle, ref, dim):
    rng = np.random.default_rng(dim)
    for _ in range(20):
        r = rng.standard_normal(dim + 1)
        A = np.array([[r[abs(i - j)] + (3.0 if i == j else 0.0) for j in range(dim)] for i in range(dim)])
        b = rng.standard_normal(dim)
        ro, xo = oracle.lesolve(A, b)
        rr, xr = ref.lesolve(A, b)
        assert ro == rr and bits_equal(xo, xr)
    ro, _ = oracle.lesolve(np.zeros((dim, dim)), np.ones(dim))
    rr, _ = ref.lesolve(np.zeros((dim, dim)), np.ones(dim))
    assert ro == rr == -1


@pytest.mark.parametrize("name", W.NAMES)
@pytest.mark.parametrize("ntaps", [1, 3, 5])
def test_longterm(oracle, ref, name, ntaps):
    res = W.gen(name, 1, 4096, 16, seed=11)[0] >> 18
    if name == "sine":
        res = (W.gen(name, 1, 4096, 16, seed=11)[0] >> 18)
    a = oracle.ltm_analyze(res, 8192, ntaps, want_autocorr=True)
    b = ref.ltm_analyze(res, 8192, ntaps, want_autocorr=True)
    assert a[0] == b[0]
    assert bits_equal(a[3], b[3])
    if a[0] == 0:
        assert a[1] == b[1] and bits_equal(a[2], b[2])


def test_longterm_pitched(oracle, ref):
    """a strongly periodic residual must pick the same pitch and taps, and filter identically"""
    rng = np.random.default_rng(3)
    period = 97
    base = rng.integers(-2000, 2000, period)
    res = (np.tile(base, 50)[:4096] + rng.integers(-50, 50, 4096)).astype(np.int32)
    for ntaps in (1, 3, 5):
        ra, pa, ca = oracle.ltm_analyze(res, 32768, ntaps)
        rb, pb, cb = ref.ltm_analyze(res, 32768, ntaps)
        assert ra == rb == 0 and pa == pb == period and bits_equal(ca, cb)
        q = (np.rint(ca * 32768).astype(np.int64) << 16).astype(np.int32)
        ya, yb = oracle.ltm_predict(res, pa, q), ref.ltm_predict(res, pb, q)
        assert np.array_equal(ya, yb)
        assert np.array_equal(oracle.ltm_synth(ya, pa, q), res)
        assert np.array_equal(ref.ltm_synth(yb, pb, q), res)


@pytest.mark.parametrize("name", W.NAMES)
@pytest.mark.parametrize("order", [4, 8, 16, 32])
def test_lms(oracle, ref, name, order):
    for bits in (16, 24):
        x = W.gen(name, 1, 8192, bits, seed=13)[0] >> (32 - bits)
        a, b = oracle.lms_predict(x, order), ref.lms_predict(x, order)
        assert np.array_equal(a, b)
        assert np.array_equal(oracle.lms_synth(a, order), x)
        assert np.array_equal(ref.lms_synth(b, order), x)
    short = W.gen(name, 1, 3, 16, seed=13)[0] >> 16
    assert np.array_equal(oracle.lms_predict(short, order), ref.lms_predict(short, order))


def test_dijkstra_known_answer(oracle, ref):
    """2-node and 7-node graphs in the style of reference test_SLAPredictor.c:807-990"""
    BIG = float(1 << 24)
    adj = np.full((2, 2), BIG)
    adj[0, 1] = 114514.0
    ro, co, po = oracle.dijkstra(adj, 0, 1)
    rr, cr, pr = ref.dijkstra(adj, 0, 1)
    assert (ro, co) == (rr, cr) == (0, 114514.0) and np.array_equal(po, pr)
    rng = np.random.default_rng(0)
    for nodes in (5, 7, 17, 30):
        adj = np.full((nodes, nodes), BIG)
        for i in range(nodes):
            for j in range(i + 1, nodes):
                if rng.random() < 0.6 or j == i + 1:
                    adj[i, j] = float(rng.integers(1, 50))
        ro, co, po = oracle.dijkstra(adj, 0, nodes - 1)
        rr, cr, pr = ref.dijkstra(adj, 0, nodes - 1)
        assert ro == rr == 0 and co == cr and np.array_equal(po, pr)


@pytest.mark.parametrize("nch", [1, 2])
@pytest.mark.parametrize("n,maxb", [(4096, 4096), (8192, 8192), (16384, 16384), (5000, 5000), (1500, 1500)])
def test_partition_search(oracle, ref, nch, n, maxb):
    pcm = W.music_like(nch, n, 16, seed=n)
    pcm[:, : n // 3] //= 16          # a level step so that splitting pays off
    x = pcm.astype(np.float64) * 2.0 ** -31
    ro, a = oracle.partition_search(x, min(2048, n), 1024, maxb, 16, 16)
    rr, b = ref.partition_search(x, min(2048, n), 1024, maxb, 16, 16)
    assert ro == rr == 0 and np.array_equal(a, b) and a.sum() == n


@pytest.mark.parametrize("name", W.NAMES)
@pytest.mark.parametrize("nch,bits", [(1, 16), (2, 24), (8, 16)])
def test_rice_coder(oracle, ref, name, nch, bits):
    res = W.gen(name, nch, 2048, bits, seed=17) >> (32 - bits + 4)
    assert np.array_equal(oracle.rice_init(res), ref.rice_init(res))
    a, b = oracle.code_residual(res, bits), ref.code_residual(res, bits)
    assert a == b
    assert np.array_equal(oracle.decode_residual(a, nch, 2048, bits), res)
    assert np.array_equal(ref.decode_residual(b, nch, 2048, bits), res)


def test_rice_coder_byte_stream(oracle, ref):
    """reference test_SLACoder.c:25-278 codes the bytes of PriChanIcon.png"""
    png = np.frombuffer(open("/root/reference/test/PriChanIcon.png", "rb").read(), np.uint8)
    res = (png.astype(np.int32) - 128)[None, :65536]
    a, b = oracle.code_residual(res, 16), ref.code_residual(res, 16)
    assert a == b
    assert np.array_equal(oracle.decode_residual(a, 1, res.shape[1], 16), res)


def test_rice_coder_escape(oracle, ref):
    """large outliers drive the quotient past 16 -> gamma escape"""
    rng = np.random.default_rng(1)
    res = rng.integers(-40, 40, (1, 4096)).astype(np.int32)
    res[0, ::97] = 3000000
    res[0, 5::211] = -2000000
    a, b = oracle.code_residual(res, 24), ref.code_residual(res, 24)
    assert a == b
    assert np.array_equal(oracle.decode_residual(a, 1, 4096, 24), res)


# --------------------------------------------------------------- codec level

PRESETS = [(8, 1, 4, 0, 0, 4096), (8, 1, 8, 1, 1, 12288), (16, 1, 8, 1, 1, 12288),
           (32, 3, 8, 1, 1, 12288), (32, 3, 8, 1, 1, 16384)]


def compare_traces(ta, tb, params):
    assert ta.num_blocks == tb.num_blocks and ta.offset_lshift == tb.offset_lshift
    nb = ta.num_blocks
    for f in ("blk_start", "blk_nsmpl", "blk_type", "blk_bytes"):
        assert np.array_equal(getattr(ta, f)[:nb], getattr(tb, f)[:nb]), f
    comp = ta.blk_type[:nb] == 0
    assert bits_equal(ta.parcor[:nb][comp], tb.parcor[:nb][comp])
    for f in ("code", "kint", "rshift", "pitch", "rice_init"):
        assert np.array_equal(getattr(ta, f)[:nb][comp], getattr(tb, f)[:nb][comp]), f
    used = ta.pitch[:nb] >= 3
    assert np.array_equal(ta.ltm_coef[:nb][used & comp[:, None]], tb.ltm_coef[:nb][used & comp[:, None]])
    for b in np.nonzero(comp)[0]:
        s, n = int(ta.blk_start[b]), int(ta.blk_nsmpl[b])
        assert np.array_equal(ta.res_lattice[:, s:s + n], tb.res_lattice[:, s:s + n])
        assert np.array_equal(ta.res_final[:, s:s + n], tb.res_final[:, s:s + n])


@pytest.mark.parametrize("preset", range(5))
def test_a_wav_presets(oracle, ref, preset):
    pcm, bits, rate = S.read_wav(A_WAV)
    po, lt, lm, ms, win, mb = PRESETS[preset]
    p = S.make_params(1, bits, rate, po, lt, lm, 0, win, mb)
    ra, da, ta = oracle.encode_trace(p, pcm)
    rb, db, tb = ref.encode_trace(p, pcm)
    assert ra == rb == 0 and da == db
    assert ref.encode_whole(p, pcm)[1] == db        # the probe's driving loop is faithful
    compare_traces(ta, tb, p)
    rd, dec, _ = oracle.decode_whole(p, da, pcm.shape[1])
    assert rd == 0 and np.array_equal(dec, pcm)
    rd, dec, _ = ref.decode_whole(p, da, pcm.shape[1])
    assert rd == 0 and np.array_equal(dec, pcm)


def test_a_wav_1024_blocks(oracle, ref):
    """BASELINE config 0 / SURVEY H7: order-8, 1024-sample EncodeBlock calls under a 2048 header"""
    pcm, bits, rate = S.read_wav(A_WAV)
    p = S.make_params(1, bits, rate, 8, 1, 4, 0, 1, 2048, cap=(1, 2048, 8, 1, 4))
    ra, da = oracle.encode_fixed_blocks(p, pcm, 1024)
    rb, db = ref.encode_fixed_blocks(p, pcm, 1024)
    assert ra == rb == 0 and da == db
    rd, dec, hdr = oracle.decode_whole(p, da, pcm.shape[1])
    assert rd == 0 and np.array_equal(dec, pcm) and hdr[9] == 235


@pytest.mark.parametrize("name", W.NAMES)
@pytest.mark.parametrize("nch", [1, 2, 8])
@pytest.mark.parametrize("bits", [8, 16, 24])
@pytest.mark.parametrize("lshift", [0, 8])
def test_roundtrip_matrix(oracle, ref, name, nch, bits, lshift):
    """reference test_SLAEncodeDecode.c:558-1172: {parcor 4, ltm 1, lms 4, SIN, 16384}"""
    if lshift >= bits:
        pytest.skip("no bits left")
    n = 8192 + 517
    pcm = W.gen(name, nch, n, bits, lshift=lshift, seed=nch * 100 + bits)
    p = S.make_params(nch, bits, 44100, 4, 1, 4, 0, 1, 16384, cap=(8, 16384, 48, 5, 40))
    ra, da, ta = oracle.encode_trace(p, pcm)
    rb, db, tb = ref.encode_trace(p, pcm)
    assert ra == rb == 0 and da == db
    compare_traces(ta, tb, p)
    rd, dec, _ = oracle.decode_whole(p, da, n)
    assert rd == 0 and np.array_equal(dec, pcm)
    rd, dec, _ = ref.decode_whole(p, da, n)
    assert rd == 0 and np.array_equal(dec, pcm)


CONFIGS = {
    # name: (nch, bits, order, ltm, lms, ms, window, max_block, capacity)
    "C2": (1, 16, 16, 1, 8, 0, 1, 4096, (1, 4096, 16, 1, 8)),
    "C3": (2, 24, 32, 3, 8, 1, 1, 4096, (2, 4096, 32, 3, 8)),
    "C4": (2, 16, 16, 1, 8, 1, 1, 4096, (2, 4096, 16, 1, 8)),
    "C5": (8, 24, 48, 3, 8, 0, 1, 8192, (8, 8192, 48, 3, 8)),
}


@pytest.mark.parametrize("cfg", sorted(CONFIGS))
@pytest.mark.parametrize("kind", ["synth", "synth_gaps", "music"])
def test_baseline_configs(oracle, ref, cfg, kind):
    nch, bits, order, ltm, lms, ms, win, mb, cap = CONFIGS[cfg]
    rate = 96000 if cfg == "C5" else 48000
    n = 40000 if cfg != "C5" else 30000
    if kind == "music":
        pcm = W.music_like(nch, n, bits, seed=5)
    else:
        pcm = S.synth_pcm(nch, n, bits, rate, gaps=(kind == "synth_gaps"))
        if kind == "synth_gaps":
            pcm[:, :3000] = 0
            pcm[:, 9000:14000] = 0
    p = S.make_params(nch, bits, rate, order, ltm, lms, ms, win, mb, cap=cap)
    ra, da, ta = oracle.encode_trace(p, pcm)
    rb, db, tb = ref.encode_trace(p, pcm)
    assert ra == rb == 0 and da == db
    compare_traces(ta, tb, p)
    rd, dec, _ = oracle.decode_whole(p, da, n)
    assert rd == 0 and np.array_equal(dec, pcm)


def test_leading_silence_block_layout(oracle, ref):
    """3000 leading zeros -> one SILENT block of exactly 3000 samples"""
    pcm = S.synth_pcm(1, 20000, 16)
    pcm[:, :3000] = 0
    p = S.make_params(1, 16, 48000, 16, 1, 8, 0, 1, 4096, cap=(1, 4096, 16, 1, 8))
    _, da, ta = oracle.encode_trace(p, pcm)
    _, db, tb = ref.encode_trace(p, pcm)
    assert da == db
    nb = ta.num_blocks
    assert list(ta.blk_start[:nb]) == [0, 3000, 7096, 11192, 15288, 19384]
    assert list(ta.blk_nsmpl[:nb]) == [3000, 4096, 4096, 4096, 4096, 616]
    assert ta.blk_type[0] == 1 and ta.blk_bytes[0] == 11


def test_raw_fallback(oracle, ref):
    """full-scale white noise is not compressible -> RAW blocks"""
    pcm = W.gen("white", 2, 9000, 16, seed=2)
    p = S.make_params(2, 16, 48000, 16, 1, 8, 1, 1, 4096, cap=(2, 4096, 16, 1, 8))
    ra, da, ta = oracle.encode_trace(p, pcm)
    rb, db, tb = ref.encode_trace(p, pcm)
    assert ra == rb == 0 and da == db
    assert (ta.blk_type[:ta.num_blocks] == 2).any()
    rd, dec, _ = oracle.decode_whole(p, da, 9000)
    assert rd == 0 and np.array_equal(dec, pcm)


@pytest.mark.parametrize("n", [1, 15, 16, 17, 100, 2047, 2048, 2049, 4096 + 15, 4096 + 17, 8192 + 1023])
def test_tiny_and_ragged_lengths(oracle, ref, n):
    """NOT covered against the reference: a tail of exactly `order` samples after a full block.
    There the reference's Levinson recursion consumes a stale autocorrelation entry
    (src/SLAPredictor.c:344-346), produces NaN costs and its Dijkstra loop never terminates
    (observed: n = 4096 + 16 at order 16 hangs the compiled reference)."""
    pcm = W.music_like(1, n, 16, seed=n)
    p = S.make_params(1, 16, 48000, 16, 1, 8, 0, 1, 4096, cap=(1, 4096, 16, 1, 8))
    ra, da = oracle.encode_whole(p, pcm)
    rb, db = ref.encode_whole(p, pcm)
    assert ra == rb == 0 and da == db
    rd, dec, _ = oracle.decode_whole(p, da, n)
    assert rd == 0 and np.array_equal(dec, pcm)


def test_tail_equal_to_order_terminates(oracle):
    pcm = W.music_like(1, 4096 + 16, 16, seed=1)
    p = S.make_params(1, 16, 48000, 16, 1, 8, 0, 1, 4096, cap=(1, 4096, 16, 1, 8))
    ret, data = oracle.encode_whole(p, pcm)
    assert ret in (0, 6)


def test_error_codes(oracle, ref):
    pcm = W.music_like(2, 5000, 16)
    bad = S.make_params(2, 16, 48000, 64, 1, 8, 0, 1, 4096)          # order over capacity 48
    assert oracle.encode_whole(bad, pcm)[0] != 0
    ms3 = S.make_params(1, 16, 48000, 8, 1, 8, 1, 1, 4096)          # MS with one channel
    ro = oracle.encode_whole(ms3, pcm[:1])[0]
    rr = ref.encode_whole(ms3, pcm[:1])[0]
    assert ro == rr == 5
    small = S.make_params(1, 16, 48000, 8, 1, 8, 0, 1, 1024)        # block below the 2048 minimum
    assert oracle.encode_whole(small, pcm[:1])[0] == 3


def test_decoder_result_codes_on_damaged_streams(oracle, ref):
    """The oracle's decoder is what the HIP decoder is checked against (tests/test_gpu_decoder.py); here its result codes
    on damaged streams are pinned to the unmodified reference decoder's.  (On failure the reference does not report how
    many samples it had already written -- src/SLADecoder.c:729 is only reached on success -- so only the codes and,
    for streams that still decode, the samples are compared.)"""
    def same(data, cap, p):
        ro, do, _ = oracle.decode_whole(p, bytes(data), cap)
        rr, dr, _ = ref.decode_whole(p, bytes(data), cap)
        assert ro == rr, (ro, rr)
        if ro == 0:
            assert np.array_equal(do, dr)
        return ro

    pcm = W.music_like(2, 30000, 16, seed=9)
    p = S.make_params(2, 16, 48000, 16, 1, 8, 1, 1, 4096)
    ret, data, tr = oracle.encode_trace(p, pcm)
    assert ret == 0
    offs = np.concatenate(([43], 43 + np.cumsum(tr.blk_bytes[:tr.num_blocks]))).astype(int)
    d = bytearray(data); d[offs[3] + 40] ^= 0x10
    assert same(d, 30000, p) == 11                       # corrupt block
    d = bytearray(data); d[20] ^= 1
    assert same(d, 30000, p) == 11                       # corrupt file header
    d = bytearray(data); d[0] = ord("X")
    assert same(d, 30000, p) == 10                       # not an SLA file
    for cut in (offs[2] + 100, offs[4], offs[4] + 5, 43, 50):
        assert same(data[:cut], 30000, p) == 9           # truncated
    d = bytearray(data); d[offs[2]] = 0x7F
    assert same(d, 30000, p) == 12                       # lost sync
    cap = int(tr.blk_start[3]) + 10
    assert same(data, cap, p) == 4                       # output buffer too small
    d = bytearray(data); d[offs[3] + 30] ^= 1
    assert same(d, cap, p) == 11                         # ... and damaged: the CRC is checked first
    small = S.make_params(2, 16, 48000, 16, 1, 8, 1, 1, 4096, cap=(2, 4096, 8, 1, 8))
    assert same(data, 30000, small) == 3                 # beyond the handle's capacity

    pcm1 = W.music_like(1, 30000, 16, seed=9)
    p1 = S.make_params(1, 16, 48000, 16, 1, 8, 0, 1, 4096)
    ret, data1, tr1 = oracle.encode_trace(p1, pcm1)
    offs1 = np.concatenate(([43], 43 + np.cumsum(tr1.blk_bytes[:tr1.num_blocks]))).astype(int)
    # a size field that disagrees with the body: the reference continues from where its bit reader stopped
    d = bytearray(data1); k = 2
    size = int.from_bytes(d[offs1[k] + 2:offs1[k] + 6], "big") + 1
    d[offs1[k] + 2:offs1[k] + 6] = size.to_bytes(4, "big")
    d[offs1[k] + 6:offs1[k] + 8] = int(oracle.crc16(bytes(d[offs1[k] + 8:offs1[k] + 6 + size]))).to_bytes(2, "big")
    assert same(d, 30000, p1) == 0
    d = bytearray(data1); d[28] = 1                      # mid/side on a mono stream
    d[8:10] = int(oracle.crc16(bytes(d[10:43]))).to_bytes(2, "big")
    assert same(d, 30000, p1) == 5
