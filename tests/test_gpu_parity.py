"""GPU parity tests (run with -m gpu on an MI355X): the HIP path, called through the C-ABI of
libsla_hip.so, must be bit-identical to the CPU oracle -- .sla bytes, block tables, PARCOR doubles
(bit patterns), transmitted codes, lattice coefficients, residual planes, Rice parameters -- on the
committed golden vectors, on seeded inputs covering the reference's own round-trip matrix
(reference test/test_SLAEncodeDecode.c:558-1172), on the edge cases (silence, ragged tails, RAW
fallback, leading-silence frame shifts) and, at BASELINE.json's full sizes, through round trips and
prefix consistency.  Nothing here reads /root/reference."""
import ctypes as C
import hashlib
import os

import numpy as np
import pytest

import slalibs as S
import waveforms as W
from test_oracle_golden import CASES as GOLDEN_CASES, check_trace_against_golden, load_case

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def hip():
    # torch (device memory for the launcher tests) must bring up its HIP runtime before
    # libsla_hip.so binds libamdhip64, otherwise torch reports "No HIP GPUs are available"
    import torch
    torch.cuda.init()
    import sla_amd
    sla_amd.lib()
    return sla_amd


def hip_encode(hip, p, pcm, want_residuals=True):
    enc = hip.Encoder(p.cap_channels, p.cap_block_samples, p.cap_parcor_order, p.cap_longterm_order, p.cap_lms_order)
    try:
        enc.set_wave_format(p.num_channels, p.bits_per_sample, p.sampling_rate)
        enc.set_encode_parameter(p.parcor_order, p.longterm_order, p.lms_order, p.ch_process_method,
                                 p.window_type, p.max_block_samples)
        enc.set_option("stream", 0)              # the analysis tables below describe an analysis on THIS handle
        data = enc.encode_whole(pcm)
        tr = enc.trace(want_residuals)
        n = pcm.shape[1]
        if n >= 4 * p.max_block_samples and n * p.num_channels <= (1 << 24):
            # the same file through the streamed path (pieces on worker lanes), in pieces as small as it takes them
            enc.set_option("stream", 1)
            enc.set_option("stream_piece", 1024)
            enc.set_option("stream_lanes", 1 + (n // 7) % 4)
            enc.set_option("tail_taps", (1, 2, 4)[(n // 5) % 3])         # the three layouts of the tail kernel take turns
            again = enc.encode_whole(pcm)
            assert again == data, "streamed EncodeWhole differs from the plain path"
        return data, tr
    finally:
        enc.close()


def assert_same_as_oracle(oracle, hip, p, pcm, roundtrip=True):
    ret, want, to = oracle.encode_trace(p, pcm)
    assert ret == 0
    got, tg = hip_encode(hip, p, pcm)
    nb = to.num_blocks
    assert tg.num_blocks == nb and tg.offset_lshift == to.offset_lshift
    for f in ("blk_start", "blk_nsmpl", "blk_type", "blk_bytes"):
        assert np.array_equal(getattr(tg, f)[:nb], getattr(to, f)[:nb]), f
    comp = to.blk_type[:nb] == 0
    assert S.parcor_same(tg, to, nb, comp)
    for f in ("code", "kint", "rshift", "pitch", "rice_init"):
        assert np.array_equal(getattr(tg, f)[:nb][comp], getattr(to, f)[:nb][comp]), f
    used = (to.pitch[:nb] >= 3) & comp[:, None]
    assert np.array_equal(tg.ltm_coef[:nb][used], to.ltm_coef[:nb][used])
    for b in np.nonzero(comp)[0]:
        s, n = int(to.blk_start[b]), int(to.blk_nsmpl[b])
        assert np.array_equal(tg.res_lattice[:, s:s + n], to.res_lattice[:, s:s + n]), ("lattice", b)
        assert np.array_equal(tg.res_final[:, s:s + n], to.res_final[:, s:s + n]), ("final", b)
    assert got == want
    if roundtrip:
        rd, dec, _ = oracle.decode_whole(p, got, pcm.shape[1])
        assert rd == 0 and np.array_equal(dec, pcm)
    return got


# ------------------------------------------------------------------ golden vectors

@pytest.mark.parametrize("name", GOLDEN_CASES)
def test_golden_vectors(hip, name):
    g, p, pcm = load_case(name)
    data, tr = hip_encode(hip, p, pcm)
    check_trace_against_golden(g, tr, data)
    sha = lambda a: hashlib.sha1(np.ascontiguousarray(a).tobytes()).hexdigest()
    comp_only = (g["blk_type"] == 0).all()
    if comp_only:
        assert sha(tr.res_lattice) == str(g["res_lattice_sha1"])
        assert sha(tr.res_final) == str(g["res_final_sha1"])


# ------------------------------------------------------------------ oracle, seeded inputs

@pytest.mark.parametrize("name", W.NAMES)
@pytest.mark.parametrize("nch", [1, 2, 8])
@pytest.mark.parametrize("bits", [8, 16, 24])
@pytest.mark.parametrize("lshift", [0, 8])
def test_roundtrip_matrix(oracle, hip, name, nch, bits, lshift):
    """the reference's own round-trip matrix: {parcor 4, ltm 1, lms 4, SIN, max block 16384}"""
    if lshift >= bits:
        pytest.skip("no bits left")
    n = 8192 + 517
    pcm = W.gen(name, nch, n, bits, lshift=lshift, seed=nch * 100 + bits)
    p = S.make_params(nch, bits, 44100, 4, 1, 4, 0, 1, 16384)
    assert_same_as_oracle(oracle, hip, p, pcm)


CONFIGS = {
    "C2": (1, 16, 48000, 16, 1, 8, 0, 1, 4096, (1, 4096, 16, 1, 8)),
    "C3": (2, 24, 48000, 32, 3, 8, 1, 1, 4096, (2, 4096, 32, 3, 8)),
    "C4": (2, 16, 48000, 16, 1, 8, 1, 1, 4096, (2, 4096, 16, 1, 8)),
    "C5": (8, 24, 96000, 48, 3, 8, 0, 1, 8192, (8, 8192, 48, 3, 8)),
}


@pytest.mark.parametrize("cfg", sorted(CONFIGS))
@pytest.mark.parametrize("kind", ["synth", "synth_gaps", "music"])
def test_baseline_configs(oracle, hip, cfg, kind):
    nch, bits, rate, order, ltm, lms, ms, win, mb, cap = CONFIGS[cfg]
    n = 150000 if cfg != "C5" else 70000
    if kind == "music":
        pcm = W.music_like(nch, n, bits, seed=5)
    else:
        pcm = S.synth_pcm(nch, n, bits, rate, gaps=(kind == "synth_gaps"))
        if kind == "synth_gaps":
            pcm[:, :3000] = 0
            pcm[:, 9000:14000] = 0
    p = S.make_params(nch, bits, rate, order, ltm, lms, ms, win, mb, cap=cap)
    assert_same_as_oracle(oracle, hip, p, pcm)


@pytest.mark.parametrize("preset", [(8, 1, 4, 0, 0, 4096), (8, 1, 8, 1, 1, 12288), (16, 1, 8, 1, 1, 12288),
                                    (32, 3, 8, 1, 1, 12288), (32, 3, 8, 1, 1, 16384)])
@pytest.mark.parametrize("nch", [1, 2])
def test_cli_presets(oracle, hip, preset, nch):
    """the reference CLI's five presets under its capacity {8,16384,48,5,40} (src/main.c:63-70,94-99)"""
    po, lt, lm, ms, win, mb = preset
    pcm = W.music_like(nch, 60000, 16, seed=nch)
    p = S.make_params(nch, 16, 44100, po, lt, lm, ms if nch == 2 else 0, win, mb)
    assert_same_as_oracle(oracle, hip, p, pcm)


@pytest.mark.parametrize("wtype", range(5))
def test_window_types(oracle, hip, wtype):
    pcm = W.music_like(1, 20000, 16, seed=wtype)
    p = S.make_params(1, 16, 48000, 16, 1, 8, 0, wtype, 4096)
    assert_same_as_oracle(oracle, hip, p, pcm)


@pytest.mark.parametrize("lms", [4, 8, 16, 32])
@pytest.mark.parametrize("ltm", [1, 3, 5])
def test_tail_orders(oracle, hip, lms, ltm):
    pcm = W.music_like(2, 20000, 24, seed=lms + ltm)
    p = S.make_params(2, 24, 48000, 16, ltm, lms, 1, 1, 4096)
    want = assert_same_as_oracle(oracle, hip, p, pcm)             # the automatic choice (few jobs: one tap per lane; order 32: two)
    got, _ = _encode_with_options(hip, p, pcm, tail_taps=1)       # one tap of each history per lane (k_tailk<., 1>; order 32 takes two)
    assert got == want
    got, _ = _encode_with_options(hip, p, pcm, tail_taps=1, tail_waves=2, chunks=1)
    assert got == want
    got, _ = _encode_with_options(hip, p, pcm, tail_taps=2)       # two (k_tailk<., 2>)
    assert got == want
    got, _ = _encode_with_options(hip, p, pcm, tail_taps=4, tail_waves=4)      # four (k_tailk<., 4>; LMS order 4: two)
    assert got == want


@pytest.mark.parametrize("taps", [1, 2, 4])
@pytest.mark.parametrize("n", [1, 7, 9, 31, 33, 63, 65, 4095, 4097, 20011])
def test_tail_k_taps_per_lane_ragged_blocks(oracle, hip, taps, n):
    """k_tailk on blocks shorter than the LMS order, one sample over a multiple of it, and with the last job of a wave
    missing (3 channels: the job count is not a multiple of the jobs per wave)"""
    pcm = W.music_like(3, n, 16, seed=n)
    p = S.make_params(3, 16, 48000, 8, 3, 8, 0, 1, 2048)
    ret, want = oracle.encode_whole(p, pcm)
    assert ret == 0
    got, _ = _encode_with_options(hip, p, pcm, tail_taps=taps)
    assert got == want


def test_pitched_signal_uses_longterm(oracle, hip):
    rng = np.random.default_rng(3)
    base = rng.integers(-6000, 6000, 131)
    x = (np.tile(base, 400)[:40000] + rng.integers(-300, 300, 40000)).astype(np.int64)
    pcm = np.ascontiguousarray((x << 16).astype(np.int32)[None, :])
    p = S.make_params(1, 16, 48000, 8, 3, 8, 0, 1, 4096)
    assert_same_as_oracle(oracle, hip, p, pcm)
    _, tr = hip_encode(hip, p, pcm, want_residuals=False)
    assert (tr.pitch[:tr.num_blocks] >= 3).any()


# ------------------------------------------------------------------ edge cases

@pytest.mark.parametrize("n", [1, 15, 17, 100, 1023, 2047, 2048, 2049, 4096, 4096 + 15, 4096 + 17, 8192 + 1023, 12345])
def test_ragged_lengths(oracle, hip, n):
    pcm = W.music_like(1, n, 16, seed=n)
    p = S.make_params(1, 16, 48000, 16, 1, 8, 0, 1, 4096, cap=(1, 4096, 16, 1, 8))
    assert_same_as_oracle(oracle, hip, p, pcm)


def test_leading_silence_shifts_frames(oracle, hip):
    """3000 leading zeros -> one SILENT block of 3000 samples, later frames shifted"""
    pcm = S.synth_pcm(1, 20000, 16)
    pcm[:, :3000] = 0
    p = S.make_params(1, 16, 48000, 16, 1, 8, 0, 1, 4096, cap=(1, 4096, 16, 1, 8))
    assert_same_as_oracle(oracle, hip, p, pcm)
    _, tr = hip_encode(hip, p, pcm, want_residuals=False)
    assert list(tr.blk_start[:tr.num_blocks]) == [0, 3000, 7096, 11192, 15288, 19384]
    assert tr.blk_type[0] == 1 and tr.blk_bytes[0] == 11


def test_all_silent_file(oracle, hip):
    pcm = np.zeros((2, 30000), np.int32)
    p = S.make_params(2, 16, 48000, 16, 1, 8, 1, 1, 4096)
    assert_same_as_oracle(oracle, hip, p, pcm)


def test_silence_inside_a_frame(oracle, hip):
    pcm = W.music_like(2, 30000, 16, seed=4)
    pcm[:, 5000:9500] = 0
    pcm[:, 20000:20900] = 0
    p = S.make_params(2, 16, 48000, 16, 1, 8, 1, 1, 4096)
    assert_same_as_oracle(oracle, hip, p, pcm)


@pytest.mark.parametrize("tail", [1, 10, 63, 64, 100, 2047])
def test_short_silent_tail_without_a_zero_mask_word(oracle, hip, tail):
    """the last super-frame may be a silent block shorter than one 64-sample mask word: the host must not
    rely on the all-zero-word count there (it only fetches the end of the mask when the count is 0)"""
    n = 3 * 4096 + tail
    pcm = W.music_like(2, n, 16, seed=tail)
    pcm[pcm == 0] = 1 << 16                 # no accidental zero samples elsewhere
    pcm[:, 3 * 4096:] = 0
    p = S.make_params(2, 16, 44100, parcor=8, ltm=1, lms=4, ms=1, max_block=4096)
    assert_same_as_oracle(oracle, hip, p, np.ascontiguousarray(pcm))


def test_raw_fallback(oracle, hip):
    pcm = W.gen("white", 2, 20000, 16, seed=2)
    p = S.make_params(2, 16, 48000, 16, 1, 8, 1, 1, 4096)
    got = assert_same_as_oracle(oracle, hip, p, pcm)
    _, tr = hip_encode(hip, p, pcm, want_residuals=False)
    assert (tr.blk_type[:tr.num_blocks] == 2).any()
    assert len(got) > 0


def test_full_scale_24bit_side_channel(oracle, hip):
    """MS on full-scale 24-bit input: 25-bit side channel -> rshift 9, wrapping lattice products"""
    rng = np.random.default_rng(9)
    l = (np.sin(np.arange(30000) * 0.05) * (2 ** 23 - 1)).astype(np.int64)
    r = -l + rng.integers(-1000, 1000, 30000)
    r = np.clip(r, -2 ** 23, 2 ** 23 - 1)
    pcm = np.ascontiguousarray((np.stack([l, r]) << 8).astype(np.int32))
    p = S.make_params(2, 24, 48000, 32, 3, 8, 1, 1, 4096)
    assert_same_as_oracle(oracle, hip, p, pcm)
    _, tr = hip_encode(hip, p, pcm, want_residuals=False)
    assert tr.rshift[:tr.num_blocks].max() >= 9


def test_offset_lshift_detected(oracle, hip):
    pcm = (W.music_like(2, 20000, 16, seed=6) >> 20) << 20      # only the top 12 bits used
    p = S.make_params(2, 16, 48000, 16, 1, 8, 0, 1, 4096)
    assert_same_as_oracle(oracle, hip, p, pcm)
    _, tr = hip_encode(hip, p, pcm, want_residuals=False)
    assert tr.offset_lshift == 4


def test_encode_block_1024_frames_a_wav(oracle, hip):
    """BASELINE config 0 (plumbing): a.wav, order 8, 1024-sample EncodeBlock calls under a header
    whose max block is 2048 -- through the same-signature SLAEncoder_EncodeBlock"""
    import os
    pcm, bits, rate = S.read_wav(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "a.wav"))
    pcm = pcm[:, :40000]
    p = S.make_params(1, bits, rate, 8, 1, 4, 0, 1, 2048, cap=(1, 2048, 8, 1, 4))
    ret, want = oracle.encode_fixed_blocks(p, pcm, 1024)
    assert ret == 0
    enc = hip.Encoder(1, 2048, 8, 1, 4)
    enc.set_wave_format(1, bits, rate)
    enc.set_encode_parameter(8, 1, 4, 0, 1, 2048)
    blocks = [enc.encode_block(pcm[:, s:s + 1024]) for s in range(0, pcm.shape[1], 1024)]
    enc.close()
    body = b"".join(blocks)
    assert body == want[43:]
    maxbps = max(8 * len(b) * rate // min(1024, pcm.shape[1] - i * 1024) for i, b in enumerate(blocks))
    hdr = hip.encode_header(1, bits, rate, 0, 8, 1, 4, 0, 1, 2048, pcm.shape[1], len(blocks),
                            max(len(b) for b in blocks), maxbps)
    assert hdr == want[:43]
    rd, dec, _ = oracle.decode_whole(p, hdr + body, pcm.shape[1])
    assert rd == 0 and np.array_equal(dec, pcm)


def test_api_error_codes(hip):
    enc = hip.Encoder(2, 4096, 16, 1, 8)
    with pytest.raises(hip.SlaError) as e:
        enc.encode_whole(np.zeros((1, 100), np.int32))
    assert e.value.code == 15                                   # PARAMETER_NOT_SET
    with pytest.raises(hip.SlaError) as e:
        enc.set_wave_format(3, 16, 48000)
    assert e.value.code == 3                                    # EXCEED_HANDLE_CAPACITY
    enc.set_wave_format(1, 16, 48000)
    with pytest.raises(hip.SlaError) as e:
        enc.set_encode_parameter(32, 1, 8, 0, 1, 4096)
    assert e.value.code == 3
    with pytest.raises(hip.SlaError) as e:
        enc.set_encode_parameter(16, 1, 8, 0, 1, 1024)
    assert e.value.code == 3
    enc.set_encode_parameter(16, 1, 8, hip.CH_STEREO_MS, 1, 4096)
    with pytest.raises(hip.SlaError) as e:
        enc.encode_whole(np.zeros((1, 5000), np.int32))
    assert e.value.code == 5                                    # INVAILD_CHPROCESSMETHOD
    enc.set_encode_parameter(16, 1, 8, 0, 1, 4096)
    with pytest.raises(hip.SlaError) as e:
        enc.encode_whole(W.music_like(1, 20000, 16), capacity=1000)
    assert e.value.code == 4                                    # INSUFFICIENT_BUFFER_SIZE
    enc.close()


def test_encoder_handle_is_reusable(oracle, hip):
    """one handle, several files and parameter changes (window pool / workspace reuse)"""
    enc = hip.Encoder(2, 8192, 32, 3, 8)
    for i, (nch, bits, order, win, mb, n) in enumerate([(1, 16, 16, 1, 4096, 30000), (2, 24, 32, 2, 8192, 50000),
                                                        (1, 16, 16, 1, 4096, 9000), (2, 16, 8, 4, 2048, 20000)]):
        pcm = W.music_like(nch, n, bits, seed=i)
        enc.set_wave_format(nch, bits, 48000)
        enc.set_encode_parameter(order, 1, 8, 0, win, mb)
        got = enc.encode_whole(pcm)
        p = S.make_params(nch, bits, 48000, order, 1, 8, 0, win, mb, cap=(2, 8192, 32, 3, 8))
        assert got == oracle.encode_whole(p, pcm)[1]
    enc.close()


def test_distinct_handles_on_distinct_threads(oracle, hip):
    """handles are not re-entrant, but distinct handles may run on distinct host threads"""
    import threading
    cases = [(1, 16, 16, 4096, 120000), (2, 24, 32, 8192, 90000), (2, 16, 8, 2048, 70000), (4, 16, 16, 4096, 60000)]
    want, pcms, bad = [], [], []
    for i, (nch, bits, order, mb, n) in enumerate(cases):
        pcm = W.music_like(nch, n, bits, seed=40 + i)
        p = S.make_params(nch, bits, 48000, parcor=order, ltm=3, lms=8, ms=int(nch == 2), max_block=mb)
        ret, data, _ = oracle.encode_trace(p, pcm)
        assert ret == 0
        want.append(data); pcms.append((p, pcm))

    def work(i):
        p, pcm = pcms[i]
        for _ in range(6):
            if hip_encode(hip, p, pcm, want_residuals=False)[0] != want[i]:
                bad.append(i)
    threads = [threading.Thread(target=work, args=(i,)) for i in range(len(cases))]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not bad


# ------------------------------------------------------------------ kernel launchers (C-ABI, torch = device memory)

def test_lattice_launcher_wraps_like_int32(oracle, hip):
    """SURVEY H4 through sla_hip_launch_lattice: full-range data and coefficients overflow int32"""
    import torch
    L = hip.lib()
    rng = np.random.default_rng(7)
    n, order = 5000, 32
    x = rng.integers(-2 ** 31, 2 ** 31 - 1, n, dtype=np.int64).astype(np.int32)
    kint = rng.integers(-32768, 32767, order + 1, dtype=np.int64).astype(np.int32)
    kint[0] = 0
    want = oracle.lattice_predict(oracle.preemph_i32(x), kint)
    L.sla_hip_lattice_chunk_samples.restype = C.c_uint32
    per = L.sla_hip_lattice_chunk_samples(order)

    class Chunk(C.Structure):
        _fields_ = [("blk_off", C.c_uint64), ("blk_len", C.c_uint32), ("chunk_start", C.c_uint32),
                    ("count", C.c_uint32), ("channel", C.c_uint32), ("slot", C.c_uint32), ("int_shift", C.c_uint32)]
    chunks = (Chunk * ((n + per - 1) // per))()
    for i in range(len(chunks)):
        chunks[i] = Chunk(0, n, i * per, min(per, n - i * per), 0, 0, 0)
    d_pcm = torch.from_numpy(x).cuda()
    d_k = torch.from_numpy(kint).cuda()
    d_chunks = torch.frombuffer(bytearray(bytes(chunks)), dtype=torch.uint8).cuda()
    d_res = torch.zeros(n, dtype=torch.int32, device="cuda")
    torch.cuda.synchronize()
    rc = L.sla_hip_launch_lattice(C.c_void_p(d_pcm.data_ptr()), C.c_uint64(n), 0, order, C.c_void_p(d_chunks.data_ptr()),
                                  len(chunks), C.c_void_p(d_k.data_ptr()), C.c_void_p(d_res.data_ptr()), None)
    assert rc == 0
    torch.cuda.synchronize()
    assert np.array_equal(d_res.cpu().numpy(), want)


def test_lpc_launcher_candidates(oracle, hip):
    """sla_hip_launch_lpc on an un-windowed window with several sub-ranges == oracle autocorr+Levinson"""
    import torch
    L = hip.lib()
    order, W_ = 16, 4096
    pcm = W.music_like(1, W_, 24, seed=12)[0]
    cand = [(0, 2048), (0, 3072), (0, 4096), (1024, 2048), (1024, 3072), (2048, 2048), (100, 37), (5, 9)]

    class Group(C.Structure):
        _fields_ = [("pcm_off", C.c_uint64)] + [(n, C.c_uint32) for n in (
            "num_samples", "channel", "win_off", "int_shift", "cand_first", "cand_count", "slot_first", "pad_")]
    groups = (Group * 1)(Group(0, W_, 0, 0xFFFFFFFF, 8, 0, len(cand), 0, 0))
    cands = np.array(cand, np.uint32)
    d_pcm = torch.from_numpy(pcm).cuda()
    d_g = torch.frombuffer(bytearray(bytes(groups)), dtype=torch.uint8).cuda()
    d_c = torch.from_numpy(cands).cuda()
    d_out = torch.zeros(len(cand) * (order + 2), dtype=torch.float64, device="cuda")
    torch.cuda.synchronize()
    rc = L.sla_hip_launch_lpc(C.c_void_p(d_pcm.data_ptr()), C.c_uint64(W_), 0, order, C.c_void_p(d_g.data_ptr()), 1,
                              W_, len(cand), C.c_void_p(d_c.data_ptr()), None, C.c_void_p(d_out.data_ptr()),
                              None, None, None, None)
    assert rc == 0
    torch.cuda.synchronize()
    out = d_out.cpu().numpy().reshape(len(cand), order + 2)
    x = pcm.astype(np.float64) * 2.0 ** -31
    for i, (s, n) in enumerate(cand):
        xs = x[s:s + n]
        r0 = oracle.autocorr(xs, 1)[0]
        _, par = oracle.parcor(xs, order)
        assert out[i, 0].hex() == r0.hex()
        assert np.array_equal(out[i, 1:].view(np.uint64), par.view(np.uint64)), (i, s, n)


@pytest.mark.parametrize("ntaps", [1, 3, 5])
def test_ltm_solve_launcher(oracle, hip, ntaps):
    """sla_hip_launch_ltm_solve (pitch + taps on the device, the long double refinement residual in integer arithmetic)
    against the host solve the CPU suite pins to the oracle (tests/test_host_logic.py::test_longterm_solve): records of
    real residuals through the oracle's own analysis, and 30000 synthetic records -- smooth, near-singular, huge, tiny,
    cancelling, refused codes, lags at both ends of the range"""
    import torch
    L = hip.lib()
    f64p, u32p = C.POINTER(C.c_double), C.POINTER(C.c_uint32)
    L.slai_ltm_solve.argtypes = [f64p, C.c_uint32, u32p, f64p]
    L.slai_ltm_solve.restype = C.c_int
    rng = np.random.default_rng(100 + ntaps)
    recs = []
    # (a) real residuals: the record the FFT kernel would hand over, from the oracle's autocorrelation
    base = rng.integers(-2000, 2000, 97)
    real = [W.gen(nm, 1, 4096, 16, seed=11)[0] >> 18 for nm in W.NAMES]
    real.append((np.tile(base, 50)[:4096] + rng.integers(-50, 50, 4096)).astype(np.int32))
    want_real = []
    for res in real:
        ret, pitch, coef, ac = oracle.ltm_analyze(np.ascontiguousarray(res, np.int32), 8192, ntaps, want_autocorr=True)
        if ret != 0 or abs(ac[0]) <= np.finfo(np.float32).tiny:
            continue
        rec = np.zeros(12)
        rec[0], rec[1] = 1.0, float(pitch)
        rec[2:7] = ac[:5]
        rec[7:12] = [ac[pitch + k - 2] if pitch + k >= 2 else 0.0 for k in range(5)]
        recs.append(rec)
        want_real.append((pitch, coef))
    assert len(want_real) >= 3
    # (b) synthetic records
    for i in range(30000):
        kind = i % 10
        rho = rng.uniform(-0.999, 0.999)
        scale = 10.0 ** rng.uniform(-6, 18) if kind != 7 else 10.0 ** rng.uniform(-40, -30)
        low = scale * rho ** np.arange(5) * (1.0 + (0 if kind == 1 else 1e-3) * rng.standard_normal(5))
        low[0] = abs(low[0]) + (0 if kind in (1, 2) else scale * rng.uniform(0, 0.5))
        if kind == 2:
            low[:] = scale                                  # rank one
        if kind == 3:
            low = scale * np.cos(0.7 * np.arange(5))        # a pure tone: singular for 3 and 5 taps
        mid = low[0] * rng.uniform(-1.2, 1.2, 5)
        if kind == 4:
            mid = low[[2, 1, 0, 1, 2]] * (1 + 1e-12 * rng.standard_normal(5))      # the taps cancel in the residual
        chosen = int(rng.integers(0, 300)) if kind == 5 else int(rng.integers(3, 256))
        code = [1.0, 1.0, 1.0, 1.0, 1.0, 1.0, 0.0, 1.0, 2.0, 1.0][kind]
        if kind == 9:
            low = np.round(low / scale * 2 ** 20) * scale / 2 ** 20      # few significant bits: exact ties in the rounding
            mid = np.round(mid / scale * 2 ** 20) * scale / 2 ** 20
        recs.append(np.concatenate([[code, float(chosen)], low, mid]))
    recs = np.ascontiguousarray(np.array(recs))
    n = len(recs)

    class Group(C.Structure):
        _fields_ = [("pcm_off", C.c_uint64)] + [(nm, C.c_uint32) for nm in (
            "num_samples", "channel", "win_off", "int_shift", "cand_first", "cand_count", "slot_first", "pad_")]

    class Job(C.Structure):
        _fields_ = [("blk_off", C.c_uint64), ("blk_len", C.c_uint32), ("channel", C.c_uint32), ("pitch", C.c_uint32),
                    ("ltm_coef", C.c_int32 * 5), ("pad_", C.c_uint32 * 2)]
    assert C.sizeof(Job) == 48
    groups = (Group * n)()
    for i in range(n):
        groups[i] = Group(1000 * i, 100 + i, i % 7, 0, 0, 0, 1, i, 0)
    d_rec = torch.from_numpy(recs).cuda()
    d_g = torch.frombuffer(bytearray(bytes(groups)), dtype=torch.uint8).cuda()
    d_jobs = torch.full((n * C.sizeof(Job),), 0xAB, dtype=torch.uint8, device="cuda")
    torch.cuda.synchronize()
    L.sla_hip_launch_ltm_solve.restype = C.c_int
    rc = L.sla_hip_launch_ltm_solve(C.c_void_p(d_rec.data_ptr()), C.c_void_p(d_g.data_ptr()), C.c_uint32(n), C.c_uint32(ntaps),
                                    C.c_void_p(d_jobs.data_ptr()), None)
    assert rc == 0
    torch.cuda.synchronize()
    jobs = (Job * n).from_buffer_copy(d_jobs.cpu().numpy().tobytes())
    solved = refused = fallback = 0
    for i in range(n):
        pitch = C.c_uint32(0)
        coef = np.zeros(5)
        ret = L.slai_ltm_solve(recs[i].ctypes.data_as(f64p), ntaps, C.byref(pitch), coef.ctypes.data_as(f64p))
        if i < len(want_real):
            assert ret == 0 and pitch.value == want_real[i][0]
            assert np.array_equal(coef[:ntaps].view(np.uint64), want_real[i][1].view(np.uint64))
        if ret != 0:
            coef[:] = 0
            refused += 1
        wp = 0 if (ret != 0 or pitch.value >= 256) else pitch.value
        q = []
        for t in range(5):
            v = (coef[t] if t < ntaps else 0.0) * 32768.0
            rv = np.floor(v + 0.5) if v >= 0 else -np.floor(-v + 0.5)
            qi = -2 ** 31 if not (-2147483649.0 < rv < 2147483648.0) else int(rv)
            q.append(((qi << 16) & 0xFFFFFFFF) - (1 << 32) if ((qi << 16) & 0x80000000) else (qi << 16) & 0xFFFFFFFF)
        j = jobs[i]
        assert (j.blk_off, j.blk_len, j.channel) == (1000 * i, 100 + i, i % 7)
        assert j.pitch == wp, (i, recs[i])
        assert list(j.ltm_coef) == q, (i, recs[i], list(j.ltm_coef), q)
        assert tuple(j.pad_) == (0, 0)
        solved += int(ret == 0 and recs[i][0] == 1.0)
        fallback += int(ret == 0 and ntaps > 1 and np.count_nonzero(coef[:ntaps]) == 1)
    assert solved > 15000 and refused > 3000
    if ntaps > 1:
        assert fallback > 100          # |taps| >= 1: the single-tap fallback was taken


@pytest.mark.parametrize("ltm", [1, 3, 5])
def test_longterm_paths_agree(oracle, hip, ltm):
    """long-term stage solved on the device (one k_tail for the file, or one per chunk) and on the host threads: the
    oracle's bytes each time -- pitched material, silence gaps, RAW blocks (white noise) in the same file"""
    rng = np.random.default_rng(40 + ltm)
    n = 300000
    base = rng.integers(-6000, 6000, 131)
    x = (np.tile(base, n // 131 + 1)[:n] + rng.integers(-300, 300, n)).astype(np.int64)
    y = W.music_like(1, n, 16, seed=ltm)[0].astype(np.int64) >> 16
    pcm = np.stack([x, y])
    pcm[:, 50000:70000] = rng.integers(-32768, 32767, (2, 20000))           # RAW blocks
    pcm[:, 120000:131000] = 0
    pcm = np.ascontiguousarray((pcm << 16).astype(np.int32))
    p = S.make_params(2, 16, 48000, parcor=8, ltm=ltm, lms=8, ms=0, max_block=4096)
    ret, want, tro = oracle.encode_trace(p, pcm)
    assert ret == 0
    assert (tro.blk_type[:tro.num_blocks] == 2).any() and (tro.pitch[:tro.num_blocks] >= 3).any()
    for opts in ({}, {"chunks": 3}, {"chunks": 3, "single_tail": 0}, {"device_ltm": 0}, {"device_ltm": 0, "chunks": 2},
                 {"device_ltm": 0, "chunks": 3, "single_tail": 0}):
        got, _ = _encode_with_options(hip, p, pcm, **opts)
        assert got == want, opts


@pytest.mark.parametrize("order,nch,bits,ms", [(16, 1, 16, 0), (32, 2, 16, 1), (5, 2, 24, 1), (48, 1, 24, 0), (10, 1, 8, 0)])
def test_search_exact_launcher_equals_serial_chains(oracle, hip, order, nch, bits, ms):
    """sla_hip_launch_search_exact (tile sums, any summation order) == oracle autocorr+Levinson in the
    reference's serial order, bit for bit, on every candidate of a ragged 16-node window; a window whose
    energy is over the limit is flagged with NaN instead"""
    import torch
    L = hip.lib()
    L.sla_hip_search_exact_lags.restype = C.c_uint32
    lags = L.sla_hip_search_exact_lags(order)
    assert lags >= order + 1
    W_ = 15 * 1024 + 333
    # quiet enough that 24-bit material stays below 2^51 units^2 over the window
    pcm = W.music_like(nch, W_ + 500, bits, seed=order)[:, 200:200 + W_ + 100]
    if bits == 24:
        pcm = (pcm >> 12) << 8
    pcm = np.ascontiguousarray(pcm)
    stride = pcm.shape[1]
    cand = []
    for i in range(17):
        for j in range(i + 1, 17):
            ln = min((j - i) * 1024, W_ - i * 1024)
            if i * 1024 < W_ and ln >= 1 and (j - i) <= 9:
                cand.append((i * 1024, ln))
    cand = sorted(set(cand))

    class Group(C.Structure):
        _fields_ = [("pcm_off", C.c_uint64)] + [(n, C.c_uint32) for n in (
            "num_samples", "channel", "win_off", "int_shift", "cand_first", "cand_count", "slot_first", "pad_")]
    off = 50
    groups = (Group * nch)(*[Group(off, W_, ch, 0xFFFFFFFF, 32 - bits, 0, len(cand), ch * len(cand), 0) for ch in range(nch)])
    d_pcm = torch.from_numpy(pcm).cuda()
    d_g = torch.frombuffer(bytearray(bytes(groups)), dtype=torch.uint8).cuda()
    d_c = torch.from_numpy(np.array(cand, np.uint32)).cuda()
    d_ts = torch.zeros(nch * 16 * 2 * lags, dtype=torch.float64, device="cuda")
    d_out = torch.zeros(nch * len(cand) * (order + 2), dtype=torch.float64, device="cuda")
    ntz = 32 - bits
    limit = 2.0 ** (51 + 2 * (ntz - 31 - ms))
    torch.cuda.synchronize()

    def run(lim):
        rc = L.sla_hip_launch_search_exact(C.c_void_p(d_pcm.data_ptr()), C.c_uint64(stride), ms, order,
                                           C.c_void_p(d_g.data_ptr()), nch, W_, len(cand), C.c_void_p(d_c.data_ptr()),
                                           C.c_void_p(d_ts.data_ptr()), C.c_void_p(d_out.data_ptr()), C.c_double(lim), C.c_double(0.0), None, None)
        assert rc == 0
        torch.cuda.synchronize()
        return d_out.cpu().numpy().reshape(nch, len(cand), order + 2)
    out = run(limit)
    x = pcm.astype(np.float64) * 2.0 ** -31
    if ms:
        x = np.stack([(x[0] + x[1]) / 2, x[0] - x[1]])
    for ch in range(nch):
        for i, (s, n) in enumerate(cand):
            xs = np.ascontiguousarray(x[ch, off + s:off + s + n])
            r0 = oracle.autocorr(xs, 1)[0]
            _, par = oracle.parcor(xs, order)
            assert out[ch, i, 0].hex() == r0.hex(), (ch, i, s, n)
            assert np.array_equal(out[ch, i, 1:].view(np.uint64), par.view(np.uint64)), (ch, i, s, n)
    assert np.isnan(run(limit * 2.0 ** -40)[:, :, 0]).all()


def _encode_with_options(hip, p, pcm, **options):
    """one whole-file encode on a fresh handle whose routes / layout knobs are set through
    sla_hip_encoder_set_option (nothing on the launch path reads the environment)"""
    enc = hip.Encoder(p.cap_channels, p.cap_block_samples, p.cap_parcor_order, p.cap_longterm_order, p.cap_lms_order)
    try:
        for k, v in options.items():
            enc.set_option(k, v)
        enc.set_wave_format(p.num_channels, p.bits_per_sample, p.sampling_rate)
        enc.set_encode_parameter(p.parcor_order, p.longterm_order, p.lms_order, p.ch_process_method,
                                 p.window_type, p.max_block_samples)
        data = enc.encode_whole(pcm)
        return data, enc.last_timing() + list(enc.last_counters())
    finally:
        enc.close()


@pytest.mark.parametrize("order,nch,bits,ms,kind", [
    (48, 1, 24, 0, "bench"), (32, 2, 24, 1, "bench"), (16, 1, 24, 0, "noise"), (32, 1, 24, 0, "tone"), (8, 2, 26, 1, "noise"),
    (48, 1, 24, 0, "mixed"), (24, 1, 32, 0, "bench"), (16, 1, 16, 0, "music")])
def test_search_certificate_brackets_the_reference(oracle, hip, order, nch, bits, ms, kind):
    """windows over the exactness limit: sla_hip_launch_search_exact keeps the tile sums and reports, per candidate, a
    half width w of log2(e_p) (e_p = r0 * prod(1 - k^2)).  The reference's value -- autocorrelation summed in ITS order,
    ITS Levinson recursion -- must lie inside [mid - w, mid + w] for every candidate: loud sinusoids + noise (the bench
    signal), full-scale noise, an almost pure tone (nearly singular), a loud window with a quiet stretch, 32-bit material;
    candidates without a bracket carry +inf.  (The limit is forced to zero so that every window takes this route.)"""
    import torch
    L = hip.lib()
    L.sla_hip_search_exact_lags.restype = C.c_uint32
    lags = L.sla_hip_search_exact_lags(order)
    W_ = 8 * 1024
    rng = np.random.default_rng(order + bits)
    n = W_ + 64
    t = np.arange(n) / 48000.0
    fs = 2.0 ** (bits - 1) - 1
    if kind == "bench":
        x = [fs * (0.35 * np.sin(2 * np.pi * 220 * (c + 1) * t) + 0.2 * np.sin(2 * np.pi * 1333.7 * t + c) + 0.1 * np.sin(2 * np.pi * 5011.3 * t))
             + rng.uniform(-0.02, 0.02, n) * fs for c in range(nch)]
    elif kind == "noise":
        x = [rng.uniform(-1, 1, n) * fs for _ in range(nch)]
    elif kind == "tone":
        x = [0.9 * fs * np.sin(2 * np.pi * 997.0 * t + c) + rng.uniform(-2, 2, n) for c in range(nch)]
    elif kind == "mixed":
        env = np.where((np.arange(n) > 3000) & (np.arange(n) < 5500), 1e-4, 1.0)
        x = [rng.uniform(-1, 1, n) * fs * env for _ in range(nch)]
    else:
        x = [W.music_like(1, n, bits, seed=3)[0].astype(np.float64) / 2.0 ** (32 - bits)]
    pcm = np.ascontiguousarray((np.round(np.stack(x)).astype(np.int64) << (32 - bits)).astype(np.int32))
    stride = pcm.shape[1]
    cand = sorted({(i * 1024, (j - i) * 1024) for i in range(9) for j in range(i + 2, 9)})

    class Group(C.Structure):
        _fields_ = [("pcm_off", C.c_uint64)] + [(k, C.c_uint32) for k in (
            "num_samples", "channel", "win_off", "int_shift", "cand_first", "cand_count", "slot_first", "pad_")]
    off = 17
    groups = (Group * nch)(*[Group(off, W_, ch, 0xFFFFFFFF, 32 - bits, 0, len(cand), ch * len(cand), 0) for ch in range(nch)])
    d_pcm = torch.from_numpy(pcm).cuda()
    d_g = torch.frombuffer(bytearray(bytes(groups)), dtype=torch.uint8).cuda()
    d_c = torch.from_numpy(np.array(cand, np.uint32)).cuda()
    d_ts = torch.zeros(nch * 16 * 2 * lags, dtype=torch.float64, device="cuda")
    d_out = torch.zeros(nch * len(cand) * (order + 2), dtype=torch.float64, device="cuda")
    torch.cuda.synchronize()
    rc = L.sla_hip_launch_search_exact(C.c_void_p(d_pcm.data_ptr()), C.c_uint64(stride), ms, order,
                                       C.c_void_p(d_g.data_ptr()), nch, W_, len(cand), C.c_void_p(d_c.data_ptr()),
                                       C.c_void_p(d_ts.data_ptr()), C.c_void_p(d_out.data_ptr()), C.c_double(0.0), C.c_double(64.0), None, None)
    assert rc == 0
    torch.cuda.synchronize()
    out = d_out.cpu().numpy().reshape(nch, len(cand), order + 2)
    xd = pcm.astype(np.float64) * 2.0 ** -31
    if ms:
        xd = np.stack([(xd[0] + xd[1]) / 2, xd[0] - xd[1]])
    bracketed, worst = 0, 0.0
    for ch in range(nch):
        for i, (s, ln) in enumerate(cand):
            xs = np.ascontiguousarray(xd[ch, off + s:off + s + ln])
            r0 = oracle.autocorr(xs, 1)[0]
            _, par = oracle.parcor(xs, order)
            o = out[ch, i]
            w = o[1]
            assert w > 0.0                                                 # marked as certified-not-exact
            assert abs(o[0] - r0) <= 1e-9 * abs(r0) + 1e-300                # the tile sums are close ...
            if not np.isfinite(w):
                continue
            bracketed += 1
            ref = np.log2(r0) + np.sum(np.log2(1.0 - par[1:] ** 2))
            mid = np.log2(o[0]) + o[2]                                      # slot of a certified candidate: { r0, w, log2(e_p / r0), 0.. }
            assert not o[3:].any()
            assert abs(ref - mid) <= w, (ch, s, ln, ref - mid, w)           # ... and the reference sits inside the bracket,
            worst = max(worst, abs(ref - mid) / w)
    assert worst < 0.05                                                     # far inside it
    if kind in ("bench", "noise", "music"):
        assert bracketed == nch * len(cand)


@pytest.mark.parametrize("nch,bits,ms,maxb", [(1, 16, 0, 4096), (2, 24, 1, 16384), (2, 16, 1, 12288)])
def test_search_paths_agree(oracle, hip, nch, bits, ms, maxb):
    """every way a partition search can run gives the oracle's bytes: tile sums (exact below the limit); every window
    pushed over a lowered limit and (a) certified by the error bracket, (b) flagged at once and rerun as serial chains
    (certificate off), (c) certificate so wide that nothing certifies -> flagged by k_plan, rerun, decided on the host;
    no tile sums at all"""
    n = 200000
    pcm = W.music_like(nch, n, bits, seed=77)
    p = S.make_params(nch, bits, 48000, parcor=16, ltm=1, lms=8, ms=ms, max_block=maxb)
    ret, want, _ = oracle.encode_trace(p, pcm)
    assert ret == 0
    a, ta = _encode_with_options(hip, p, pcm)
    b, tb = _encode_with_options(hip, p, pcm, exact_bits=1)
    b1, tb1 = _encode_with_options(hip, p, pcm, exact_bits=1, cert_safety=0)
    b2, tb2 = _encode_with_options(hip, p, pcm, exact_bits=1, cert_safety=1e25)
    b3, tb3 = _encode_with_options(hip, p, pcm, exact_bits=1, device_plan=0)
    c, tc = _encode_with_options(hip, p, pcm, search_exact=0)
    assert a == want and b == want and b1 == want and b2 == want and b3 == want and c == want
    assert ta[11] == 1.0 and tb[11] == 1.0 and tc[11] == 0.0
    assert tb1[10] > 0 and tb2[10] == tb1[10] and tb3[10] == tb1[10] and tc[10] == 0
    assert tb[10] <= tb1[10] // 10                 # certified: (almost) nothing had to take the chains
    if bits <= 16:
        assert ta[10] == 0          # 16-bit material can never reach the limit


@pytest.mark.parametrize("nch,bits,ms,maxb", [(1, 16, 0, 4096), (2, 24, 1, 16384), (8, 16, 0, 8192)])
def test_plan_paths_agree(oracle, hip, nch, bits, ms, maxb):
    """partition decided on the device (certified), on the host for every super-frame (margin raised so that
    nothing certifies) and with the device plan switched off: the oracle's bytes each time"""
    n = 150000
    pcm = W.music_like(nch, n, bits, seed=91) >> (4 if bits == 24 else 0)
    if bits == 24:
        pcm = (pcm >> 8) << 8
    pcm = np.ascontiguousarray(pcm)
    p = S.make_params(nch, bits, 48000, parcor=16, ltm=1, lms=8, ms=ms, max_block=maxb)
    ret, want, _ = oracle.encode_trace(p, pcm)
    assert ret == 0
    a, ta = _encode_with_options(hip, p, pcm)
    b, tb = _encode_with_options(hip, p, pcm, plan_margin=1e30)
    c, tc = _encode_with_options(hip, p, pcm, device_plan=0)
    assert a == want and b == want and c == want
    assert ta[15] == 1 and tb[15] == 1 and tc[15] == 0
    assert tb[13] > 0 and tb[13] >= ta[13] and tc[13] == 0
    assert ta[13] <= 2          # music: practically every super-frame certifies


@pytest.mark.parametrize("nch,bits,order,maxb", [(1, 16, 16, 4096), (2, 24, 32, 8192), (3, 16, 48, 16384), (2, 8, 5, 2048)])
def test_lattice_fused_into_the_block_kernel(oracle, hip, nch, bits, order, maxb):
    """option fuse_lattice: sla_hip_launch_lpc_blocks runs the PARCOR lattice inside k_lpc_blocks; option
    lpc_blocks_chains: the chosen blocks take k_lpc's serial chains"""
    pcm = W.music_like(nch, 90000, bits, seed=order)
    p = S.make_params(nch, bits, 48000, parcor=order, ltm=3, lms=8, ms=int(nch == 2), max_block=maxb)
    ret, want, _ = oracle.encode_trace(p, pcm)
    assert ret == 0
    got, _ = _encode_with_options(hip, p, pcm, fuse_lattice=1)
    assert got == want
    got, _ = _encode_with_options(hip, p, pcm, lpc_blocks_chains=1, lpc_pack=1, tail_waves=2)
    assert got == want


def test_search_loud_24bit_takes_the_fallback_where_needed(oracle, hip):
    """full-scale 24-bit noise has > 2^53 units^2 per window: certified partitions, or (certificate off) serial chains"""
    rng = np.random.default_rng(5)
    n = 70000
    loud = rng.integers(-2 ** 23, 2 ** 23 - 1, (2, n), dtype=np.int64).astype(np.int32)
    loud[:, 30000:] >>= 9           # second half quiet: stays on the tile-sum path
    pcm = np.ascontiguousarray(loud << 8)
    p = S.make_params(2, 24, 96000, parcor=32, ltm=3, lms=16, ms=1, max_block=8192)
    ret, want, _ = oracle.encode_trace(p, pcm)
    assert ret == 0
    got, t = _encode_with_options(hip, p, pcm)
    assert got == want
    assert t[11] == 1.0
    certified_run = t[10]
    got, t = _encode_with_options(hip, p, pcm, cert_safety=0)
    assert got == want and t[10] >= 8 and certified_run <= t[10] // 4       # certified: the chains are the exception


# ------------------------------------------------------------------ seeded random walk over the parameter space

def _fuzz_case(seed):
    rng = np.random.default_rng(1000 + seed)
    nch = int(rng.choice([1, 1, 2, 2, 3, 6, 8]))
    bits = int(rng.choice([8, 16, 16, 24, 24]))
    order = int(rng.choice([1, 2, 5, 8, 16, 24, 32, 48]))
    ltm = int(rng.choice([1, 3, 5]))
    lms = int(rng.choice([4, 8, 16, 32]))
    ms = int(nch == 2 and rng.integers(0, 2))
    win = int(rng.integers(0, 5))
    maxb = int(rng.choice([1024, 2048, 3000, 4096, 8192, 12288, 16384]))
    n = int(rng.choice([1, 63, 700, 2047, 2048, 4097, 9999, 20000, 33333, 50000]))
    kind = rng.choice(["music", "loud", "quiet", "gaps", "tail_zero", "wave"])
    if kind == "wave":
        pcm = W.gen(str(rng.choice(W.NAMES)), nch, n, bits, seed=seed)
    else:
        pcm = W.music_like(nch, n, bits, seed=seed, level=1.0 if kind == "loud" else (0.01 if kind == "quiet" else 0.5))
        if kind == "gaps" and n > 5000:
            a = int(rng.integers(0, n // 2)); b = a + int(rng.integers(1, n // 2))
            pcm[:, a:b] = 0
        if kind == "tail_zero":
            pcm[:, n - int(rng.integers(1, max(2, min(n, 3000)))):] = 0
    p = S.make_params(nch, bits, int(rng.choice([44100, 48000, 96000])), parcor=order, ltm=ltm, lms=lms, ms=ms, window=win,
                      max_block=maxb)
    return p, np.ascontiguousarray(pcm)


@pytest.mark.parametrize("seed", range(int(os.environ.get("SLA_FUZZ_CASES", "48"))))
def test_random_parameter_walk(oracle, hip, seed):
    """random (seeded) combinations of channels, widths, orders, windows, block limits, lengths and content
    -- silence gaps, silent tails, full-scale and very quiet material -- through every fast path and its fallback"""
    p, pcm = _fuzz_case(seed)
    ret = oracle.encode_trace(p, pcm)[0]
    if ret == 6 and pcm.shape[1] % p.max_block_samples == p.parcor_order:
        # a tail of exactly `order` samples sends the reference into a loop that never ends (the
        # oracle gives up with FAILED_TO_CALCULATE_COEF); the HIP path terminates -- nothing to compare
        pytest.skip("reference does not terminate on this input")
    if ret != 0:
        # the reference refuses the combination (e.g. max block below the minimum block of the search): same code
        with pytest.raises(hip.SlaError) as err:
            hip_encode(hip, p, pcm)
        assert err.value.code == ret
        return
    # bytes, tables and residuals against the oracle; the PCM round trip is the reference's own property and does not hold
    # for every corner the walk reaches (full-scale mid/side overflows in the reference too)
    assert_same_as_oracle(oracle, hip, p, pcm, roundtrip=False)


@pytest.mark.parametrize("seed", range(int(os.environ.get("SLA_FUZZ_LONG_CASES", "10"))))
def test_random_parameter_walk_chunked(oracle, hip, seed):
    """the same walk on files long enough for the three-chunk pipeline (>= 64 super-frames), with silence in them"""
    rng = np.random.default_rng(5000 + seed)
    nch = int(rng.choice([1, 2, 2, 4]))
    bits = int(rng.choice([16, 16, 24]))
    order = int(rng.choice([4, 16, 32]))
    maxb = int(rng.choice([2048, 4096, 8192]))
    n = int(maxb * rng.integers(70, 130) + rng.integers(0, maxb))
    pcm = W.music_like(nch, n, bits, seed=seed, level=float(rng.choice([0.02, 0.5, 1.0])))
    for _ in range(int(rng.integers(0, 4))):
        a = int(rng.integers(0, n - 1)); b = min(n, a + int(rng.integers(1, 6 * maxb)))
        pcm[:, a:b] = 0
    p = S.make_params(nch, bits, 48000, parcor=order, ltm=int(rng.choice([1, 3])), lms=int(rng.choice([8, 16])),
                      ms=int(nch == 2 and rng.integers(0, 2)), window=int(rng.integers(0, 5)), max_block=maxb)
    pcm = np.ascontiguousarray(pcm)
    assert_same_as_oracle(oracle, hip, p, pcm, roundtrip=False)
    want = oracle.encode_trace(p, pcm)[1]
    for opts in ({"chunks": 3}, {"chunks": 2, "single_tail": 0}, {"chunks": 3, "device_ltm": 0}, {"chunks": 2, "alt_streams": 1},
                 {"chunks": 3, "alt_streams": 1}, {"chunks": 2, "device_expand": 0}, {"chunks": 1, "tail_taps": 2},
                 {"chunks": 2, "tail_taps": 4, "tail_waves": 2}):
        got, t = _encode_with_options(hip, p, pcm, **opts)
        assert got == want and t[9] == opts["chunks"], opts


# ------------------------------------------------------------------ streamed SLAEncoder_EncodeWhole

def _streamed(hip, p, pcm, piece, lanes, capacity=None, **options):
    enc = hip.Encoder(p.cap_channels, p.cap_block_samples, p.cap_parcor_order, p.cap_longterm_order, p.cap_lms_order)
    try:
        enc.set_wave_format(p.num_channels, p.bits_per_sample, p.sampling_rate)
        enc.set_encode_parameter(p.parcor_order, p.longterm_order, p.lms_order, p.ch_process_method,
                                 p.window_type, p.max_block_samples)
        enc.set_option("stream_piece", piece)
        enc.set_option("stream_lanes", lanes)
        for k, v in options.items():
            enc.set_option(k, v)
        first = enc.encode_whole(pcm, capacity=capacity)
        second = enc.encode_whole(pcm, capacity=capacity)         # the lanes are reused
        assert first == second
        return first
    finally:
        enc.close()


@pytest.mark.parametrize("nch,bits,ms,maxb,piece,lanes", [
    (1, 16, 0, 4096, 1024, 1), (1, 16, 0, 4096, 20000, 2), (2, 16, 1, 4096, 40000, 3), (2, 24, 1, 2048, 9000, 4),
    (8, 24, 0, 8192, 300000, 3), (3, 16, 0, 16384, 1024, 2)])
def test_streamed_encode_whole(oracle, hip, nch, bits, ms, maxb, piece, lanes):
    """a file long enough for several pieces: upload, analysis, pack and download of different pieces overlap on the
    worker lanes; the bytes are the oracle's.  Silence runs lie across piece borders (the hop must carry over), start
    pieces, fill whole pieces"""
    rng = np.random.default_rng(nch * 100 + lanes)
    n = 40 * maxb + 777
    pcm = W.music_like(nch, n, bits, seed=piece % 97)
    per = max((piece // nch + 63) // 64 * 64, (2 * maxb + 63) // 64 * 64)
    for k in range(1, n // per + 1):
        if k % 3 == 1:
            pcm[:, k * per - 1500:k * per + 900] = 0              # across the border
        elif k % 3 == 2:
            pcm[:, k * per:k * per + int(rng.integers(1, 3 * maxb))] = 0
    if n > 5 * per:
        pcm[:, 3 * per - 10:4 * per + 10] = 0                     # a whole piece and more
    pcm = np.ascontiguousarray(pcm)
    p = S.make_params(nch, bits, 48000, parcor=16, ltm=3, lms=8, ms=ms, max_block=maxb)
    ret, want, _ = oracle.encode_trace(p, pcm)
    assert ret == 0
    assert _streamed(hip, p, pcm, piece, lanes) == want
    assert _streamed(hip, p, pcm, piece, 6) == want               # six lanes, pieces to whichever is free


def test_streamed_pieces_disagree_on_offset_lshift(oracle, hip):
    """offset_lshift belongs to the whole file: piece 0 speaks for it.  Later samples with lower bits set, or a silent
    piece 0, send the file down the plain path -- the oracle's bytes either way"""
    maxb, n = 4096, 30 * 4096
    p = S.make_params(2, 24, 48000, parcor=8, ltm=1, lms=8, ms=0, max_block=maxb)
    base = W.music_like(2, n, 24, seed=5)
    coarse = (base >> 12) << 12                                   # 20 significant bits ...
    mixed = coarse.copy()
    mixed[:, 20 * maxb:] = base[:, 20 * maxb:]                    # ... then all 24 from the 20th block on
    late = np.zeros_like(base)
    late[:, 9 * maxb:] = base[:, 9 * maxb:]                       # piece 0 is silent
    finer_first = base.copy()
    finer_first[:, 10 * maxb:] = coarse[:, 10 * maxb:]            # piece 0 has the lowest bit: later pieces agree with it
    for pcm in (mixed, late, finer_first, coarse):
        pcm = np.ascontiguousarray(pcm)
        ret, want, _ = oracle.encode_trace(p, pcm)
        assert ret == 0
        assert _streamed(hip, p, pcm, 16384, 3) == want


def test_streamed_buffer_too_small(oracle, hip):
    pcm = W.music_like(2, 30 * 4096, 16, seed=8)
    p = S.make_params(2, 16, 48000, parcor=8, ltm=1, lms=8, ms=1, max_block=4096)
    ret, want, _ = oracle.encode_trace(p, pcm)
    assert ret == 0
    for cap in (len(want) - 1, len(want) // 2, 43, 44 + 100):
        with pytest.raises(hip.SlaError) as err:
            _streamed(hip, p, pcm, 16384, 3, capacity=cap)
        assert err.value.code == 4, cap                            # SLA_APIRESULT_INSUFFICIENT_BUFFER_SIZE
    assert _streamed(hip, p, pcm, 16384, 3, capacity=len(want)) == want


# ------------------------------------------------------------------ BASELINE sizes: size-independent properties

def _full_size_check(oracle, hip, p, pcm, prefix_frames):
    got, tr = hip_encode(hip, p, pcm, want_residuals=False)
    n = pcm.shape[1]
    nb = tr.num_blocks
    assert int(tr.blk_nsmpl[:nb].sum()) == n and (tr.blk_start[1:nb] == np.cumsum(tr.blk_nsmpl[:nb - 1])).all()
    assert len(got) == 43 + int(tr.blk_bytes[:nb].sum())
    # (1) encode -> decode round trip to the original PCM (oracle decoder checks every block CRC16)
    rd, dec, hdr = oracle.decode_whole(p, got, n)
    assert rd == 0 and hdr[9] == nb and np.array_equal(dec, pcm)
    # (2) prefix consistency: blocks are independent, so the oracle's encode of the first frames
    #     must reproduce the first bytes of the big file exactly
    m = prefix_frames * p.max_block_samples
    ret, want, to = oracle.encode_trace(p, np.ascontiguousarray(pcm[:, :m]))
    assert ret == 0 and to.offset_lshift == tr.offset_lshift
    assert want[43:] == got[43:len(want)]
    assert S.parcor_same(tr, to, to.num_blocks)


def test_full_size_c2(oracle, hip):
    """BASELINE config 1: 48 kHz 16-bit mono, 10 min, order 16, 4096-sample frames"""
    pcm = S.synth_pcm(1, 48000 * 600, 16, 48000)
    p = S.make_params(1, 16, 48000, 16, 1, 8, 0, 1, 4096, cap=(1, 4096, 16, 1, 8))
    _full_size_check(oracle, hip, p, pcm, 40)


def test_c3_shape_five_minutes(oracle, hip):
    """BASELINE config 2 shape (48 kHz 24-bit stereo, order 32, MS), 5 of its 60 minutes"""
    pcm = S.synth_pcm(2, 48000 * 300, 24, 48000)
    p = S.make_params(2, 24, 48000, 32, 3, 8, 1, 1, 4096, cap=(2, 4096, 32, 3, 8))
    _full_size_check(oracle, hip, p, pcm, 20)


def test_c5_shape_one_minute(oracle, hip):
    """BASELINE config 4 shape (96 kHz 24-bit 8 channels, order 48, 8192 frames), 1 of its 30 minutes"""
    pcm = S.synth_pcm(8, 96000 * 60, 24, 96000)
    p = S.make_params(8, 24, 96000, 48, 3, 8, 0, 1, 8192, cap=(8, 8192, 48, 3, 8))
    _full_size_check(oracle, hip, p, pcm, 6)


def test_c4_batch_of_clips(oracle, hip):
    """BASELINE config 3 shape: a batch of independent stereo clips through one encoder handle"""
    enc = hip.Encoder(2, 4096, 16, 1, 8)
    enc.set_wave_format(2, 16, 48000)
    enc.set_encode_parameter(16, 1, 8, hip.CH_STEREO_MS, 1, 4096)
    p = S.make_params(2, 16, 48000, 16, 1, 8, 1, 1, 4096, cap=(2, 4096, 16, 1, 8))
    for seed in range(12):
        pcm = S.synth_pcm(2, 48000, 16, 48000, seed=1000 + seed)
        got = enc.encode_whole(pcm)
        assert got == oracle.encode_whole(p, pcm)[1]
    enc.close()


@pytest.mark.parametrize("kind,nch,bits,ms", [("music", 1, 16, 0), ("music", 2, 24, 1), ("white", 2, 16, 1),
                                               ("gaps", 2, 16, 1), ("quiet", 1, 16, 0), ("spiky", 2, 16, 0)])
def test_host_and_device_pack_agree(oracle, hip, kind, nch, bits, ms):
    """the same analysis packed by the host threads (sla_hip_pack) and by the device kernels
    (sla_hip_pack_device: k_rice_len / k_rice_write / k_block_crc) gives identical bytes == oracle;
    covers adaptive recursive Rice, fixed Golomb (quiet), gamma escapes (spiky), RAW and SILENT blocks"""
    import torch
    n = 40000
    if kind == "music":
        pcm = W.music_like(nch, n, bits, seed=21)
    elif kind == "white":
        pcm = W.gen("white", nch, n, bits, seed=3)
    elif kind == "gaps":
        pcm = S.synth_pcm(nch, n, bits, gaps=True)
        pcm[:, :3000] = 0
    elif kind == "quiet":
        rng = np.random.default_rng(5)
        pcm = (rng.integers(-3, 4, (nch, n)).astype(np.int64) << 16).astype(np.int32)
        pcm[:, ::977] = 9000 << 16                       # rare outliers in a tiny-residual block: long unary runs
    else:
        rng = np.random.default_rng(6)
        x = (W.music_like(nch, n, bits, seed=2).astype(np.int64) >> 16) // 64
        x[:, ::53] += rng.integers(-30000, 30000, x[:, ::53].shape)
        pcm = (np.clip(x, -32768, 32767) << 16).astype(np.int32)
    pcm = np.ascontiguousarray(pcm)
    p = S.make_params(nch, bits, 48000, 16, 1, 8, ms, 1, 4096)
    want = oracle.encode_whole(p, pcm)[1]
    enc = hip.Encoder()
    enc.set_wave_format(nch, bits, 48000)
    enc.set_encode_parameter(16, 1, 8, ms, 1, 4096)
    stride = (n + 63) // 64 * 64
    d = torch.zeros((nch, stride), dtype=torch.int32, device="cuda")
    d[:, :n] = torch.from_numpy(pcm).cuda()
    torch.cuda.synchronize()
    enc.analyze_device(d.data_ptr(), stride, n)
    cap = 8 * nch * n + 65536
    host_bytes = enc.pack(cap, on_device=False)
    dev_bytes = enc.pack(cap, on_device=True)
    enc.close()
    assert host_bytes == want
    assert dev_bytes == want


# ------------------------------------------------------------------ BASELINE configs 2 and 4 at their FULL length
# 60 min of 24-bit stereo and 30 min of 24-bit 8-channel audio: the input is synthesised on the device (bench.py's
# generator; seconds instead of the minutes numpy needs), encoded through SLAEncoder_EncodeWhole from host memory and
# decoded again through SLADecoder_DecodeWhole -- the round trip must return the input; the oracle pins the bytes of the
# first super-frames (blocks are independent) and the header's totals are checked.  SLA_SKIP_FULL_SIZE=1 skips them.

def _full_length_check(oracle, hip, p, cap, nch, n, bits, rate, prefix_frames):
    import time
    import torch
    import bench
    t0 = time.time()
    pcm = np.ascontiguousarray(bench.synth_device(torch, nch, n, bits, rate, 0, n).cpu().numpy())
    torch.cuda.empty_cache()
    t1 = time.time()
    enc = hip.Encoder(*cap)
    enc.set_wave_format(nch, bits, rate)
    enc.set_encode_parameter(p.parcor_order, p.longterm_order, p.lms_order, p.ch_process_method, p.window_type, p.max_block_samples)
    out = np.zeros(min(4 * nch * n + 65536, 0xFFFFFFF0), np.uint8)      # the API's sizes are 32-bit
    data = enc.encode_whole(pcm, out=out)                               # streamed: pieces on worker lanes
    t2 = time.time()
    enc.set_option("stream", 0)
    plain = enc.encode_whole(pcm, out=np.zeros(len(data) + 65536, np.uint8))
    t2b = time.time()
    assert len(plain) == len(data) and np.array_equal(plain, data)
    del plain
    enc.close()
    print("streamed %.3f s, plain %.3f s" % (t2 - t1, t2b - t2))
    dec = hip.Decoder(cap[0], cap[1], cap[2], cap[3], cap[4])
    rc, back = dec.decode_whole(data, n)
    t3 = time.time()
    dec.close()
    assert rc == 0 and back.shape == pcm.shape and np.array_equal(back, pcm)
    m = prefix_frames * p.max_block_samples
    ret, want, to = oracle.encode_trace(p, np.ascontiguousarray(pcm[:, :m]))
    keep = 43 + int(to.blk_bytes[:to.num_blocks - 1].sum())               # (the prefix's last block ends the oracle's "file")
    assert ret == 0 and bytes(data[43:keep]) == want[43:keep]
    rch, h = hip.decode_header(data[:43])
    assert rch == 0 and h.num_samples == n
    # two more windows pinned to the oracle, in the middle and at the end of the file: walk the block
    # headers (sync 0xFFFF, 32-bit size of what follows, CRC16, 16-bit sample count: src/SLAEncoder.c:685-693) to find the
    # bytes of the blocks that start at a super-frame start; the oracle encodes the same samples as a range of this file
    offs, starts, pos, off = [], [], 0, 43
    raw = data if isinstance(data, np.ndarray) else np.frombuffer(data, np.uint8)
    while off < len(raw):
        assert raw[off] == 0xFF and raw[off + 1] == 0xFF
        size = (int(raw[off + 2]) << 24) | (int(raw[off + 3]) << 16) | (int(raw[off + 4]) << 8) | int(raw[off + 5])
        offs.append(off); starts.append(pos)
        pos += (int(raw[off + 8]) << 8) | int(raw[off + 9])
        off += 6 + size
    assert pos == n and len(offs) == h.num_blocks
    offs.append(off)
    starts = np.array(starts, np.int64)
    maxb = p.max_block_samples
    lshift = int(to.offset_lshift)
    for s0 in ((n // 2) // maxb * maxb, (n // maxb - prefix_frames) * maxb):
        b0 = int(np.searchsorted(starts, s0))
        assert starts[b0] == s0                                          # no silence in this signal: super-frames sit on the block grid
        mm = min(prefix_frames * maxb, n - s0) if s0 + prefix_frames * maxb < n - maxb else n - s0
        retw, wantw = oracle.encode_range(p, np.ascontiguousarray(pcm[:, s0:s0 + mm]), lshift)
        to_end = (s0 + mm == n)
        b1 = b0
        ends = np.append(starts[1:], n)
        while b1 < len(starts) and (to_end or ends[b1] <= s0 + mm - maxb):      # (the window's last super-frame ends the oracle's "file")
            b1 += 1
        seg = bytes(raw[offs[b0]:offs[b1]])
        assert retw == 0 and b1 > b0 and seg == wantw[43:43 + len(seg)] and (not to_end or len(wantw) == 43 + len(seg)), (s0, b0, b1)
    print("full length: %d ch x %d samples, synth %.1f s, encode %.2f s (%.0f Msamples/s end to end), decode %.2f s, %d bytes, %d blocks"
          % (nch, n, t1 - t0, t2 - t1, nch * n / (t2 - t1) / 1e6, t3 - t2, len(data), h.num_blocks))


@pytest.mark.skipif(os.environ.get("SLA_SKIP_FULL_SIZE") == "1", reason="SLA_SKIP_FULL_SIZE=1")
def test_c3_full_sixty_minutes(oracle, hip):
    p = S.make_params(2, 24, 48000, 32, 3, 8, 1, 1, 4096, cap=(2, 4096, 32, 3, 8))
    _full_length_check(oracle, hip, p, (2, 4096, 32, 3, 8), 2, 48000 * 3600, 24, 48000, 20)


@pytest.mark.skipif(os.environ.get("SLA_SKIP_FULL_SIZE") == "1", reason="SLA_SKIP_FULL_SIZE=1")
def test_c5_full_thirty_minutes(oracle, hip):
    p = S.make_params(8, 24, 96000, 48, 3, 8, 0, 1, 8192, cap=(8, 8192, 48, 3, 8))
    _full_length_check(oracle, hip, p, (8, 8192, 48, 3, 8), 8, 96000 * 1800, 24, 96000, 6)


@pytest.mark.parametrize("nch,bits,n", [(2, 24, 300001), (1, 20, 70000), (3, 24, 4099)])
def test_upload24_gives_the_same_bytes(oracle, hip, nch, bits, n):
    """option "upload24": pageable input of <= 24 significant bits crosses the bus as three bytes per sample and is
    re-expanded by k_unpack24 -- same planes on the device, same file"""
    pcm = W.music_like(nch, n, bits, seed=n % 97)
    p = S.make_params(nch, bits, 48000, 16, 1, 8, 0, 1, 4096)
    ret, want = oracle.encode_whole(p, pcm)
    assert ret == 0
    got, _ = _encode_with_options(hip, p, pcm, upload24=1)
    assert got == want
    got, _ = _encode_with_options(hip, p, pcm, upload24=1, stream=1, stream_piece=65536)
    assert got == want


@pytest.mark.parametrize("n", [1, 63, 64, 65, 127, 129, 4095, 4097, 20011, 131072 + 77])
@pytest.mark.parametrize("kind", ["music", "quiet", "gaps"])
def test_rice_walk_kernels_agree(oracle, hip, n, kind):
    """the Rice parameter walk of the device pack as one lane per job (k_rice_k) and as the two-lane pipeline (k_rice_k2):
    same bytes as the oracle from both, on ragged block lengths (batches of 64 samples: one under / over), job counts that
    are not a multiple of the eight jobs of a wave (3 channels), fixed-Golomb blocks (quiet material) and silent ones"""
    pcm = W.music_like(3, n, 16, seed=n % 89, level=0.004 if kind == "quiet" else 0.5)
    if kind == "gaps" and n > 9000:
        pcm[:, 3000:8000] = 0
    p = S.make_params(3, 16, 48000, 8, 1, 8, 0, 1, 2048)
    ret, want = oracle.encode_whole(p, pcm)
    assert ret == 0
    for lanes in (1, 2):
        got, _ = _encode_with_options(hip, p, pcm, rice_lanes=lanes, stream=0)
        assert got == want, lanes


@pytest.mark.parametrize("n,bits,ms", [(8192 * 6 + 333, 24, 0), (8192 * 3 + 40, 16, 0), (4096 * 9 + 1999, 24, 1), (1500, 24, 0)])
def test_tile_sums_at_52_lags(oracle, hip, n, bits, ms):
    """orders 33 .. 52 take the tile sums from k_acf_tiles_lds<13> (partners from LDS, tile walked backwards): the oracle's bytes
    on windows with a short last tile (fewer samples than lags in it), a window shorter than a tile, loud 24-bit material
    (certified sums) and 16-bit (exact sums)"""
    nch = 2 if ms else 1
    pcm = W.music_like(nch, n, bits, seed=n % 71, level=1.0)
    p = S.make_params(nch, bits, 96000, 48, 3, 8, ms, 1, 8192)
    ret, want = oracle.encode_whole(p, pcm)
    assert ret == 0
    got, _ = _encode_with_options(hip, p, pcm, stream=0)
    assert got == want
