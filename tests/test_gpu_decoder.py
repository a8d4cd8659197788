"""GPU parity tests of the decode path (run with -m gpu on an MI355X).

SLADecoder_DecodeWhole of libsla_hip.so must return the same result code and the same samples as the CPU
oracle's decoder (oracle/sla_oracle.c: slao_decode_whole, pinned against the unmodified reference decoder -- samples and the
result codes on damaged streams -- in tests/test_oracle_vs_ref.py) on streams written by the oracle's encoder: the reference's own round-trip matrix
(test/test_SLAEncodeDecode.c:558-1172), BASELINE.json's configurations, every filter order the kernels
specialise on, silent / raw blocks, ragged lengths, and damaged streams (the first failing block decides the
code; the samples before it are delivered).  Nothing here reads /root/reference."""
import numpy as np
import pytest

import slalibs as S
import waveforms as W

pytestmark = pytest.mark.gpu

OK, BUF, CHPROC, SYNTH, DATA, HDRFMT, CORRUPT, SYNC = 0, 4, 5, 8, 9, 10, 11, 12


@pytest.fixture(scope="module")
def hip():
    import torch
    torch.cuda.init()
    import sla_amd
    sla_amd.lib()
    return sla_amd


def hip_decode(hip, p, data, capacity, crc=1):
    dec = hip.Decoder(p.cap_channels, p.cap_block_samples, p.cap_parcor_order, p.cap_longterm_order,
                      min(p.cap_lms_order, 32), crc)
    try:
        return dec.decode_whole(data, capacity, num_channels=max(int(p.num_channels), 1))
    finally:
        dec.close()


def assert_decodes_like_oracle(oracle, hip, p, data, capacity, expect=None):
    ro, want, _ = oracle.decode_whole(p, data, capacity)
    rg, got = hip_decode(hip, p, data, capacity)
    assert rg == ro, (rg, ro)
    if expect is not None:
        assert rg == expect
    assert got.shape[1] == want.shape[1]
    n = min(got.shape[0], want.shape[0])
    assert np.array_equal(got[:n], want[:n])
    return rg, got


def encode(oracle, p, pcm):
    ret, data = oracle.encode_whole(p, pcm)
    assert ret == 0
    return data


# ------------------------------------------------------------------ the reference's round-trip matrix

@pytest.mark.parametrize("name", W.NAMES)
@pytest.mark.parametrize("nch", [1, 2, 8])
@pytest.mark.parametrize("bits", [8, 16, 24])
@pytest.mark.parametrize("lshift", [0, 8])
def test_roundtrip_matrix(oracle, hip, name, nch, bits, lshift):
    if lshift >= bits:
        pytest.skip("no bits left")
    n = 8192 + 517
    pcm = W.gen(name, nch, n, bits, lshift=lshift, seed=nch * 100 + bits)
    p = S.make_params(nch, bits, 44100, 4, 1, 4, 0, 1, 16384)
    data = encode(oracle, p, pcm)
    rc, got = assert_decodes_like_oracle(oracle, hip, p, data, n, expect=OK)
    assert np.array_equal(got, pcm)


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
    data = encode(oracle, p, pcm)
    rc, got = assert_decodes_like_oracle(oracle, hip, p, data, n, expect=OK)
    assert np.array_equal(got, pcm)


@pytest.mark.parametrize("lms", [4, 8, 16, 32])
@pytest.mark.parametrize("ltm", [1, 3, 5])
def test_filter_orders(oracle, hip, lms, ltm):
    pcm = W.music_like(2, 20000, 24, seed=lms + ltm)
    p = S.make_params(2, 24, 48000, 16, ltm, lms, 1, 1, 4096)
    data = encode(oracle, p, pcm)
    rc, got = assert_decodes_like_oracle(oracle, hip, p, data, 20000, expect=OK)
    assert np.array_equal(got, pcm)


@pytest.mark.parametrize("order", [1, 3, 4, 15, 16, 17, 31, 32, 33, 48, 64, 65, 100, 128, 129, 200, 255])
def test_parcor_orders(oracle, hip, order):
    """every lattice specialisation: 16 / 32 / 64 lanes per block, 1 / 2 / 4 stages per lane"""
    pcm = W.music_like(2, 12000, 16, seed=order)
    p = S.make_params(2, 16, 48000, order, 1, 8, 0, 1, 4096, cap=(2, 4096, 255, 1, 8))
    data = encode(oracle, p, pcm)
    rc, got = assert_decodes_like_oracle(oracle, hip, p, data, 12000, expect=OK)
    assert np.array_equal(got, pcm)


def test_pitched_signal_uses_longterm(oracle, hip):
    rng = np.random.default_rng(3)
    for period, taps in ((131, 3), (37, 5), (7, 1), (5, 5), (255, 3)):
        base = rng.integers(-6000, 6000, period)
        x = (np.tile(base, 40000 // period + 1)[:40000] + rng.integers(-300, 300, 40000)).astype(np.int64)
        pcm = np.ascontiguousarray((x << 16).astype(np.int32)[None, :])
        p = S.make_params(1, 16, 48000, 8, taps, 8, 0, 1, 4096)
        ret, data, tr = oracle.encode_trace(p, pcm)
        assert ret == 0
        rc, got = assert_decodes_like_oracle(oracle, hip, p, data, 40000, expect=OK)
        assert np.array_equal(got, pcm)
        if period in (131, 37):
            assert (tr.pitch[:tr.num_blocks] >= 3).any()


@pytest.mark.parametrize("n", [1, 3, 15, 17, 100, 1023, 2047, 2048, 2049, 4096, 4096 + 15, 8192 + 1023, 12345])
def test_ragged_lengths(oracle, hip, n):
    pcm = W.music_like(1, n, 16, seed=n)
    p = S.make_params(1, 16, 48000, 16, 1, 8, 0, 1, 4096, cap=(1, 4096, 16, 1, 8))
    data = encode(oracle, p, pcm)
    rc, got = assert_decodes_like_oracle(oracle, hip, p, data, n, expect=OK)
    assert np.array_equal(got, pcm)


def test_raw_silent_and_compressed_blocks_in_one_stream(oracle, hip):
    rng = np.random.default_rng(11)
    n = 40000
    pcm = W.music_like(2, n, 16, seed=2)
    pcm[:, 5000:12000] = 0                                                 # silent blocks
    pcm[:, 20000:30000] = (rng.integers(-32768, 32768, (2, 10000)) << 16).astype(np.int32)    # incompressible -> RAW
    for ms in (0, 1):
        p = S.make_params(2, 16, 48000, 16, 1, 8, ms, 1, 4096)
        ret, data, tr = oracle.encode_trace(p, pcm)
        assert ret == 0
        types = set(int(t) for t in tr.blk_type[:tr.num_blocks])
        assert types == {0, 1, 2}, types
        rc, got = assert_decodes_like_oracle(oracle, hip, p, data, n, expect=OK)
        assert np.array_equal(got, pcm)


def test_small_residual_uses_fixed_golomb(oracle, hip):
    """mean folded residual <= 8 -> the stateless Golomb branch of the coder (src/SLACoder.c:443-450)"""
    rng = np.random.default_rng(5)
    pcm = (rng.integers(-3, 4, (2, 20000)) << 24).astype(np.int32)
    p = S.make_params(2, 8, 48000, 8, 1, 4, 0, 1, 4096)
    ret, data, tr = oracle.encode_trace(p, pcm)
    assert ret == 0 and (tr.rice_init[:tr.num_blocks] <= 8).all()
    rc, got = assert_decodes_like_oracle(oracle, hip, p, data, 20000, expect=OK)
    assert np.array_equal(got, pcm)


def test_loud_signal_exercises_the_gamma_escape(oracle, hip):
    """spikes on a quiet floor: quotients >= 16 take the gamma escape (src/SLACoder.c:248-262)"""
    rng = np.random.default_rng(6)
    x = rng.integers(-40, 40, (1, 30000)).astype(np.int64)
    x[0, ::97] = rng.integers(-2 ** 22, 2 ** 22, x[0, ::97].shape)
    pcm = (x << 8).astype(np.int32)
    p = S.make_params(1, 24, 48000, 8, 1, 8, 0, 1, 4096)
    data = encode(oracle, p, pcm)
    rc, got = assert_decodes_like_oracle(oracle, hip, p, data, 30000, expect=OK)
    assert np.array_equal(got, pcm)


# ------------------------------------------------------------------ damaged streams: first failing block decides

def _stream(oracle, nch=2, n=30000, ms=1):
    pcm = W.music_like(nch, n, 16, seed=9)
    p = S.make_params(nch, 16, 48000, 16, 1, 8, ms if nch == 2 else 0, 1, 4096)
    ret, data, tr = oracle.encode_trace(p, pcm)
    assert ret == 0
    offs = np.concatenate(([43], 43 + np.cumsum(tr.blk_bytes[:tr.num_blocks]))).astype(int)
    return p, pcm, bytearray(data), offs, tr


def test_corrupt_block_is_detected_and_earlier_blocks_are_delivered(oracle, hip):
    p, pcm, data, offs, tr = _stream(oracle)
    data[offs[3] + 40] ^= 0x10
    rc, got = assert_decodes_like_oracle(oracle, hip, p, bytes(data), pcm.shape[1], expect=CORRUPT)
    done = int(tr.blk_start[3])
    assert got.shape[1] == done and np.array_equal(got, pcm[:, :done])


def test_corrupt_file_header(oracle, hip):
    p, pcm, data, offs, tr = _stream(oracle)
    data[20] ^= 1
    assert_decodes_like_oracle(oracle, hip, p, bytes(data), pcm.shape[1], expect=CORRUPT)
    data2 = bytearray(bytes(data)); data2[0] = ord("X")
    assert_decodes_like_oracle(oracle, hip, p, bytes(data2), pcm.shape[1], expect=HDRFMT)


def test_truncated_stream(oracle, hip):
    p, pcm, data, offs, tr = _stream(oracle)
    for cut in (offs[2] + 100, offs[4], offs[4] + 5, 43, 50):
        rc, got = assert_decodes_like_oracle(oracle, hip, p, bytes(data[:cut]), pcm.shape[1], expect=DATA)
    rg, got = hip_decode(hip, p, bytes(data[:20]), pcm.shape[1])
    assert rg == DATA


def test_lost_sync(oracle, hip):
    p, pcm, data, offs, tr = _stream(oracle)
    data[offs[2]] = 0x7F
    rc, got = assert_decodes_like_oracle(oracle, hip, p, bytes(data), pcm.shape[1], expect=SYNC)
    assert got.shape[1] == int(tr.blk_start[2])


def test_output_buffer_too_small(oracle, hip):
    p, pcm, data, offs, tr = _stream(oracle)
    cap = int(tr.blk_start[3]) + 10
    rc, got = assert_decodes_like_oracle(oracle, hip, p, bytes(data), cap, expect=BUF)
    assert got.shape[1] == int(tr.blk_start[3])
    # a damaged block that also does not fit reports the damage (the CRC is checked first)
    data[offs[3] + 30] ^= 0x01
    assert_decodes_like_oracle(oracle, hip, p, bytes(data), cap, expect=CORRUPT)


def test_size_field_that_disagrees_with_the_body(oracle, hip):
    """the reference continues from where its bit reader stopped, not from the size field (src/SLADecoder.c:715):
    a size field one byte too large (CRC re-made so that only this inconsistency remains) loses sync at the
    NEXT block, after the block itself has been delivered"""
    p, pcm, data, offs, tr = _stream(oracle, nch=1, ms=0)
    k = 2
    size = int.from_bytes(data[offs[k] + 2:offs[k] + 6], "big") + 1
    data[offs[k] + 2:offs[k] + 6] = size.to_bytes(4, "big")
    crc = oracle.crc16(np.frombuffer(bytes(data[offs[k] + 8:offs[k] + 6 + size]), np.uint8))
    data[offs[k] + 6:offs[k] + 8] = int(crc).to_bytes(2, "big")
    rc, got = assert_decodes_like_oracle(oracle, hip, p, bytes(data), pcm.shape[1])
    assert got.shape[1] == int(tr.blk_start[k + 1]) or rc == OK


def test_mid_side_needs_two_channels(oracle, hip):
    p, pcm, data, offs, tr = _stream(oracle, nch=1, ms=0)
    data[28] = 1                                            # ch_process_method = MS on a mono stream
    crc = oracle.crc16(np.frombuffer(bytes(data[10:43]), np.uint8))
    data[8:10] = int(crc).to_bytes(2, "big")
    assert_decodes_like_oracle(oracle, hip, p, bytes(data), pcm.shape[1], expect=CHPROC)


def test_capacity_is_enforced(oracle, hip):
    p, pcm, data, offs, tr = _stream(oracle)
    small = S.make_params(2, 16, 48000, 16, 1, 8, 1, 1, 4096, cap=(2, 4096, 8, 1, 8))       # parcor 16 > 8
    assert_decodes_like_oracle(oracle, hip, small, bytes(data), pcm.shape[1], expect=3)
    mono = S.make_params(1, 16, 48000, 16, 1, 8, 0, 1, 4096, cap=(1, 4096, 16, 1, 8))
    rg, _ = hip_decode(hip, mono, bytes(data), pcm.shape[1])
    assert rg == 3


def test_argument_checks(hip):
    import ctypes as C
    L = hip.lib()
    assert L.SLADecoder_Create(None) is None
    h = hip.SLAHeaderInfo()
    assert L.SLADecoder_DecodeHeader(None, 43, C.byref(h)) == 2
    dec = hip.Decoder()
    try:
        n = C.c_uint32(0)
        assert L.SLADecoder_DecodeWhole(dec._h, None, 0, None, 0, C.byref(n)) == 2
        assert L.SLADecoder_SetWaveFormat(dec._h, None) == 2
        assert L.SLADecoder_SetEncodeParameter(None, None) == 2
    finally:
        dec.close()
    for bad in (dict(max_num_channels=9), dict(max_num_block_samples=32768), dict(max_longterm_order=7),
                dict(max_parcor_order=256)):
        with pytest.raises(RuntimeError):
            hip.Decoder(**bad)


def test_header_fields(oracle, hip):
    p, pcm, data, offs, tr = _stream(oracle)
    rc, h = hip.decode_header(bytes(data))
    assert rc == 0
    assert (h.wave_format.num_channels, h.wave_format.bit_per_sample, h.wave_format.sampling_rate) == (2, 16, 48000)
    assert (h.encode_param.parcor_order, h.encode_param.longterm_order, h.encode_param.lms_order_per_filter) == (16, 1, 8)
    assert h.encode_param.ch_process_method == 1 and h.encode_param.max_num_block_samples == 4096
    assert h.num_samples == pcm.shape[1] and h.num_blocks == tr.num_blocks
    assert h.max_block_size == int(tr.blk_bytes[:tr.num_blocks].max())


def test_handle_is_reusable_and_crc_check_can_be_switched_off(oracle, hip):
    p, pcm, data, offs, tr = _stream(oracle)
    dec = hip.Decoder(enable_crc_check=0)
    try:
        for _ in range(3):
            rc, got = dec.decode_whole(bytes(data), pcm.shape[1])
            assert rc == 0 and np.array_equal(got, pcm)
        # a flipped CRC field goes unnoticed without the check, the samples are untouched
        bad = bytearray(bytes(data)); bad[offs[1] + 6] ^= 0xFF
        rc, got = dec.decode_whole(bytes(bad), pcm.shape[1])
        assert rc == 0 and np.array_equal(got, pcm)
        short = W.music_like(1, 5000, 16, seed=1)
        p1 = S.make_params(1, 16, 48000, 8, 1, 4, 0, 1, 4096)
        rc, got = dec.decode_whole(encode(oracle, p1, short), 5000)
        assert rc == 0 and np.array_equal(got, short)
    finally:
        dec.close()


# ------------------------------------------------------------------ streaming decoder

@pytest.mark.parametrize("cfg", [(1, 16, 48000, 16, 1, 8, 0, 4096), (2, 24, 44100, 32, 3, 8, 1, 12288), (8, 16, 96000, 8, 1, 4, 0, 2048)])
@pytest.mark.parametrize("hz", [120.0, 30.0, 1000.0])
def test_streaming_decoder_fed_like_the_reference_cli(oracle, hip, cfg, hz):
    """src/main.c:277-420: first the largest block's worth of bytes, then the estimate per call -> the PCM of DecodeWhole"""
    nch, bits, rate, order, ltm, lms, ms, mb = cfg
    n = 40000
    pcm = W.music_like(nch, n, bits, seed=int(hz))
    pcm[:, 9000:13000] = 0                                # a silent block on the way
    p = S.make_params(nch, bits, rate, order, ltm, lms, ms, 1, mb)
    data = encode(oracle, p, pcm)
    rc, got, calls = hip.streaming_decode(data, decode_interval_hz=hz)
    assert rc == 0 and np.array_equal(got, pcm)
    per_call = int(np.ceil(np.float32(1.05) * np.float32(rate) / np.float32(hz)))
    assert calls >= -(-n // per_call)


@pytest.mark.parametrize("chunk", [1, 7, 64, 1000, 100000])
def test_streaming_decoder_with_arbitrary_fragments(oracle, hip, chunk):
    pcm = W.music_like(2, 20000, 16, seed=chunk)
    p = S.make_params(2, 16, 48000, 16, 1, 8, 1, 1, 4096)
    data = encode(oracle, p, pcm)
    rc, got, calls = hip.streaming_decode(data, feed=lambda i: chunk)
    assert rc == 0 and np.array_equal(got, pcm)


def test_streaming_decoder_bookkeeping_and_errors(oracle, hip):
    import ctypes as C
    L = hip.lib()
    pcm = W.music_like(1, 12000, 16, seed=3)
    p = S.make_params(1, 16, 48000, 8, 1, 4, 0, 1, 4096)
    data = np.frombuffer(encode(oracle, p, pcm), np.uint8)
    rc, h = hip.decode_header(data)
    u8p, u32p, i32p = hip.u8p, hip.u32p, hip.i32p
    assert L.SLAStreamingDecoder_Create(None) is None
    bad = hip.SLAStreamingDecoderConfig(hip.SLADecoderConfig(1, 4096, 8, 1, 4, 1, 0), 0.0, 16)
    assert L.SLAStreamingDecoder_Create(C.byref(bad)) is None
    cfg = hip.SLAStreamingDecoderConfig(hip.SLADecoderConfig(1, 4096, 8, 1, 4, 1, 0), 100.0, 16)
    dec = L.SLAStreamingDecoder_Create(C.byref(cfg))
    assert dec
    try:
        out = np.zeros(20000, np.int32)
        ptrs = (i32p * 1)(out.ctypes.data_as(i32p))
        got, v = C.c_uint32(0), C.c_uint32(0)
        assert L.SLAStreamingDecoder_Decode(dec, ptrs, 20000, C.byref(got)) == 15             # parameters not set
        wf24 = hip.SLAWaveFormat(1, 24, 48000, 0)
        assert L.SLAStreamingDecoder_SetWaveFormat(dec, C.byref(wf24)) == 3                 # deeper than max_bit_per_sample
        assert L.SLAStreamingDecoder_SetWaveFormat(dec, C.byref(h.wave_format)) == 0
        assert L.SLAStreamingDecoder_SetEncodeParameter(dec, C.byref(h.encode_param)) == 0
        assert L.SLAStreamingDecoder_GetOutputNumSamplesPerDecode(dec, C.byref(v)) == 0 and v.value == 504   # ceil(1.05 * 48000 / 100)
        assert L.SLAStreamingDecoder_EstimateMinimumNessesaryDataSize(dec, C.byref(v)) == 0 and v.value == 504 * 2   # 1 ch x 16 bit before any block
        assert L.SLAStreamingDecoder_Decode(dec, ptrs, 20000, C.byref(got)) == 9              # nothing appended yet
        body = data[43:]
        # nine fragments: the ninth does not fit the queue of eight
        for i in range(8):
            assert L.SLAStreamingDecoder_AppendDataFragment(dec, body[i * 10:].ctypes.data_as(u8p), 10) == 0
        assert L.SLAStreamingDecoder_GetRemainDataSize(dec, C.byref(v)) == 0 and v.value == 80
        assert L.SLAStreamingDecoder_AppendDataFragment(dec, body[80:].ctypes.data_as(u8p), 10) == 3
        ptr, size = u8p(), C.c_uint32(0)
        for i in range(8):
            assert L.SLAStreamingDecoder_CollectDataFragment(dec, C.byref(ptr), C.byref(size)) == 0 and size.value == 10
        assert L.SLAStreamingDecoder_CollectDataFragment(dec, C.byref(ptr), C.byref(size)) == 14   # no data fragments
        assert L.SLAStreamingDecoder_Decode(dec, ptrs, 20000, C.byref(got)) == 0 and got.value == 0   # a block header, not yet the block
        assert L.SLAStreamingDecoder_AppendDataFragment(dec, body[80:].ctypes.data_as(u8p), len(body) - 80) == 0
        done = 0
        while done < 12000:
            ptrs = (i32p * 1)(out[done:].ctypes.data_as(i32p))
            assert L.SLAStreamingDecoder_Decode(dec, ptrs, 12000 - done, C.byref(got)) == 0
            assert got.value == min(504, 12000 - done)
            done += got.value
        assert np.array_equal(out[:12000], pcm[0])
        assert L.SLAStreamingDecoder_EstimateDecodableNumSamples(dec, C.byref(v)) == 0 and v.value == 0
        junk = np.full(64, 0x55, np.uint8)
        assert L.SLAStreamingDecoder_AppendDataFragment(dec, junk.ctypes.data_as(u8p), 64) == 0
        assert L.SLAStreamingDecoder_Decode(dec, ptrs, 100, C.byref(got)) == 12            # no sync code
        assert L.SLAStreamingDecoder_Decode(None, ptrs, 100, C.byref(got)) == 2
    finally:
        L.SLAStreamingDecoder_Destroy(dec)


# ------------------------------------------------------------------ the codec end to end on the device

def test_encode_then_decode_on_the_device_full_c2(hip):
    """BASELINE C2 at full size (28.8 M samples): SLAEncoder_EncodeWhole -> SLADecoder_DecodeWhole is the identity"""
    n = 48000 * 600
    pcm = S.synth_pcm(1, n, 16, 48000)
    enc = hip.Encoder(1, 4096, 16, 1, 8)
    try:
        enc.set_wave_format(1, 16, 48000)
        enc.set_encode_parameter(16, 1, 8, 0, 1, 4096)
        data = enc.encode_whole(pcm)
    finally:
        enc.close()
    dec = hip.Decoder(1, 4096, 16, 1, 8)
    try:
        rc, got = dec.decode_whole(data, n)
        assert rc == 0 and got.shape == pcm.shape
        assert np.array_equal(got, pcm)
    finally:
        dec.close()


def test_random_parameter_walk(oracle, hip):
    """seeded fuzz over channels, depths, orders, block sizes and signal kinds"""
    import os
    rng = np.random.default_rng(20261003)
    cases = int(os.environ.get("SLA_FUZZ_CASES", "32"))
    for i in range(cases):
        nch = int(rng.choice([1, 2, 3, 8]))
        bits = int(rng.choice([8, 16, 24]))
        order = int(rng.choice([2, 4, 8, 10, 16, 24, 32, 48]))
        ltm = int(rng.choice([1, 3, 5]))
        lms = int(rng.choice([4, 8, 16, 32]))
        mb = int(rng.choice([2048, 3072, 4096, 8192, 16384]))
        ms = int(rng.integers(0, 2)) if nch == 2 else 0
        n = int(rng.integers(1, 30000))
        kind = rng.choice(["music", "synth", "gaps", "noise"])
        if kind == "music":
            pcm = W.music_like(nch, n, bits, seed=i, level=float(rng.uniform(0.05, 0.7)))
        elif kind == "noise":
            pcm = W.gen("white", nch, n, bits, seed=i)
        else:
            pcm = S.synth_pcm(nch, n, bits, 48000, seed=i + 1, gaps=(kind == "gaps"))
        p = S.make_params(nch, bits, 48000, order, ltm, lms, ms, int(rng.integers(0, 5)), mb)
        ret, data = oracle.encode_whole(p, pcm)
        if ret != 0:
            continue
        ctx = (i, nch, bits, order, ltm, lms, mb, ms, n, str(kind))
        ro, want, _ = oracle.decode_whole(p, data, n)
        rg, got = hip_decode(hip, p, data, n)
        assert rg == ro, ctx
        assert np.array_equal(got, want), ctx


def test_garbage_reaches_the_kernels_without_harm(oracle, hip):
    """CRC check off, bytes damaged at random (headers, coefficients, entropy-coded bodies, size fields): whatever the
    result code, every call returns (every reader loop is bounded by the stream length), nothing is written outside the
    caller's buffers, and the handle decodes a clean stream afterwards"""
    rng = np.random.default_rng(99)
    pcm = W.music_like(2, 30000, 16, seed=12)
    p = S.make_params(2, 16, 48000, 16, 3, 8, 1, 1, 4096)
    clean = encode(oracle, p, pcm)
    dec = hip.Decoder(2, 4096, 16, 3, 8, enable_crc_check=0)
    try:
        seen = set()
        for trial in range(150):
            data = bytearray(clean)
            kind = trial % 5
            if kind == 0:                                   # a few bit flips anywhere behind the file header
                for _ in range(int(rng.integers(1, 6))):
                    data[int(rng.integers(43, len(data)))] ^= 1 << int(rng.integers(0, 8))
            elif kind == 1:                                 # a burst of random bytes
                a = int(rng.integers(43, len(data) - 300))
                data[a:a + 256] = rng.integers(0, 256, 256, dtype=np.uint8).tobytes()
            elif kind == 2:                                 # zeros: endless unary runs
                a = int(rng.integers(43, len(data) - 3000))
                data[a:a + 2048] = bytes(2048)
            elif kind == 3:                                 # truncated somewhere
                data = data[:int(rng.integers(43, len(data)))]
            else:                                           # ones: maximal codes
                a = int(rng.integers(43, len(data) - 3000))
                data[a:a + 1024] = b"\xff" * 1024
            guard = np.full((2, 30000 + 64), 0x5A5A5A5A, np.int32)
            import ctypes as C
            ptrs = (hip.i32p * 2)(guard[0].ctypes.data_as(hip.i32p), guard[1].ctypes.data_as(hip.i32p))
            n = C.c_uint32(0)
            buf = np.frombuffer(bytes(data), np.uint8)
            rc = hip.lib().SLADecoder_DecodeWhole(dec._h, buf.ctypes.data_as(hip.u8p), len(buf), ptrs, 30000, C.byref(n))
            seen.add(rc)
            assert 0 <= rc <= 15 and n.value <= 30000
            assert (guard[:, 30000:] == 0x5A5A5A5A).all()
        rc, got = dec.decode_whole(clean, 30000)
        assert rc == 0 and np.array_equal(got, pcm)
        assert len(seen) >= 2, seen
    finally:
        dec.close()
