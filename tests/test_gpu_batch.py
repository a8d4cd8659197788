"""GPU parity tests of sla_hip_encode_batch (BASELINE C4: a batch of clips in one pass; run with -m gpu).

Every file of a batch must come out byte-identical to the oracle's encode of that file alone -- own header, own
offset_lshift, own super-frame grid, silent / raw / compressed blocks, ragged tails -- and to SLAEncoder_EncodeWhole of
the same handle.  Nothing here reads /root/reference."""
import numpy as np
import pytest

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


def make_encoder(hip, p):
    enc = hip.Encoder(p.cap_channels, p.cap_block_samples, p.cap_parcor_order, p.cap_longterm_order, p.cap_lms_order)
    enc.set_wave_format(p.num_channels, p.bits_per_sample, p.sampling_rate)
    enc.set_encode_parameter(p.parcor_order, p.longterm_order, p.lms_order, p.ch_process_method, p.window_type, p.max_block_samples)
    return enc


def check_batch(oracle, hip, p, pcms, also_single=True):
    enc = make_encoder(hip, p)
    try:
        got = enc.encode_batch(pcms)
        assert len(got) == len(pcms)
        for i, (pcm, (rc, data)) in enumerate(zip(pcms, got)):
            ret, want = oracle.encode_whole(p, pcm)
            assert ret == 0
            assert rc == 0, (i, rc)
            assert data == want, ("file", i, len(data), len(want))
        if also_single:                       # the same handle keeps working file by file, before and after a batch
            assert enc.encode_whole(pcms[0]) == got[0][1]
            again = enc.encode_batch(pcms[:2])
            assert [d for _, d in again] == [d for _, d in got[:2]]
    finally:
        enc.close()


def test_c4_clips(oracle, hip):
    """48 kHz 16-bit stereo MS, order 16, 4096-sample frames: clips of different lengths and contents"""
    p = S.make_params(2, 16, 48000, 16, 1, 8, 1, 1, 4096, cap=(2, 4096, 16, 1, 8))
    lens = [48000, 30001, 4096, 4097, 1, 2047, 2048, 100000, 777, 12288]
    pcms = [S.synth_pcm(2, n, 16, 48000, seed=50 + i) if i % 2 == 0 else W.music_like(2, n, 16, seed=i) for i, n in enumerate(lens)]
    check_batch(oracle, hip, p, pcms)


def test_files_with_silence_raw_and_different_lshift(oracle, hip):
    rng = np.random.default_rng(4)
    n = 30000
    a = W.music_like(2, n, 16, seed=1)
    a[:, :5000] = 0                                     # leading silence: SILENT block, shifted frames
    b = (rng.integers(-32768, 32768, (2, n)) << 16).astype(np.int32)      # incompressible: RAW blocks
    c = (W.music_like(2, n, 16, seed=2) >> 18) << 18    # two low bits of the 16 are always zero: offset_lshift 2
    d = np.zeros((2, 9000), np.int32)                   # all silent
    e = W.music_like(2, n, 16, seed=3)
    e[:, -3000:] = 0                                    # trailing silence up to the end of the file
    f = (W.music_like(2, 5000, 16, seed=4) >> 20) << 20 # offset_lshift 4
    p = S.make_params(2, 16, 48000, 16, 1, 8, 1, 1, 4096)
    check_batch(oracle, hip, p, [a, b, c, d, e, f])


@pytest.mark.parametrize("cfg", [(1, 16, 16, 1, 8, 0, 4096), (2, 24, 32, 3, 8, 1, 4096), (8, 24, 48, 3, 8, 0, 8192),
                                 (1, 8, 4, 1, 4, 0, 16384), (3, 24, 10, 5, 32, 0, 3072)])
def test_other_formats(oracle, hip, cfg):
    nch, bits, order, ltm, lms, ms, mb = cfg
    p = S.make_params(nch, bits, 48000, order, ltm, lms, ms, 1, mb)
    pcms = [W.music_like(nch, n, bits, seed=n) for n in (20000, 3000, 8193)] + [S.synth_pcm(nch, 15000, bits, 48000, seed=9, gaps=True)]
    check_batch(oracle, hip, p, pcms, also_single=False)


def test_buffer_too_small_is_reported_per_file(oracle, hip):
    p = S.make_params(2, 16, 48000, 16, 1, 8, 1, 1, 4096)
    pcms = [W.music_like(2, 20000, 16, seed=i) for i in range(3)]
    enc = make_encoder(hip, p)
    try:
        got = enc.encode_batch(pcms, capacities=[200000, 1000, 200000])
        assert [rc for rc, _ in got] == [0, 4, 0]
        for i in (0, 2):
            assert got[i][1] == oracle.encode_whole(p, pcms[i])[1]
        got = enc.encode_batch(pcms, capacities=[200000, 20, 200000])
        assert [rc for rc, _ in got] == [0, 4, 0]
    finally:
        enc.close()


def test_many_small_files(oracle, hip):
    """200 files, lengths 1 .. 6000, mono 16-bit"""
    rng = np.random.default_rng(8)
    p = S.make_params(1, 16, 48000, 8, 1, 4, 0, 1, 4096)
    pcms = [W.music_like(1, int(n), 16, seed=int(n)) for n in rng.integers(1, 6000, 200)]
    check_batch(oracle, hip, p, pcms, also_single=False)


def test_argument_checks(hip):
    import ctypes as C
    L = hip.lib()
    p = S.make_params(1, 16, 48000, 8, 1, 4, 0, 1, 4096)
    enc = make_encoder(hip, p)
    try:
        assert L.sla_hip_encode_batch(None, None, 0) == 2
        assert L.sla_hip_encode_batch(enc._h, None, 3) == 2
        assert L.sla_hip_encode_batch(enc._h, None, 0) == 0
        items = (hip.BatchItem * 1)()
        assert L.sla_hip_encode_batch(enc._h, items, 1) == 2          # NULL planes
    finally:
        enc.close()
    raw = hip.Encoder()
    try:
        items = (hip.BatchItem * 1)()
        assert L.sla_hip_encode_batch(raw._h, items, 1) == 15         # parameters not set
    finally:
        raw.close()


def test_batch_analysis_on_device_resident_planes(oracle, hip):
    """sla_hip_analyze_batch_device: files back to back in HBM, one pipeline pass; the block table, PARCOR doubles,
    codes, Rice parameters and residual planes of every file equal the oracle's analysis of that file alone"""
    import torch
    p = S.make_params(2, 16, 48000, 16, 1, 8, 1, 1, 4096, cap=(2, 4096, 16, 1, 8))
    lens = [30000, 4097, 12288, 1, 20000]
    pcms = [W.music_like(2, n, 16, seed=70 + i) for i, n in enumerate(lens)]
    pcms[2][:, :5000] = 0
    starts, at = [], 0
    for n in lens:
        starts.append(at)
        at += (n + 1023) // 1024 * 1024
    span = at
    d_pcm = torch.zeros((2, span), dtype=torch.int32, device="cuda")
    for s0, x in zip(starts, pcms):
        d_pcm[:, s0:s0 + x.shape[1]] = torch.from_numpy(x).cuda()
    torch.cuda.synchronize()
    enc = make_encoder(hip, p)
    try:
        timing, lsh = enc.analyze_batch_device(d_pcm.data_ptr(), span, span, starts, lens)
        assert list(lsh) == [0] * len(lens)
        enc.num_channels, enc.order, enc.ltm_order = 2, 16, 1
        tr = enc.trace()
        nb = tr.num_blocks
        b = 0
        for s0, x in zip(starts, pcms):
            ret, _, to = oracle.encode_trace(p, x)
            assert ret == 0
            k = to.num_blocks
            assert np.array_equal(tr.blk_start[b:b + k], to.blk_start[:k] + s0)
            assert np.array_equal(tr.blk_nsmpl[b:b + k], to.blk_nsmpl[:k]) and np.array_equal(tr.blk_type[b:b + k], to.blk_type[:k])
            comp = to.blk_type[:k] == 0
            assert S.parcor_same(tr, to, k, comp, gslice=slice(b, b + k))
            for f in ("code", "kint", "rshift", "pitch", "rice_init"):
                assert np.array_equal(getattr(tr, f)[b:b + k][comp], getattr(to, f)[:k][comp]), f
            for j in np.nonzero(comp)[0]:
                a, n = int(to.blk_start[j]), int(to.blk_nsmpl[j])
                assert np.array_equal(tr.res_final[:, s0 + a:s0 + a + n], to.res_final[:, a:a + n])
            b += k
        assert b == nb
        # a start off the 1024-sample grid is refused
        with pytest.raises(hip.SlaError):
            enc.analyze_batch_device(d_pcm.data_ptr(), span, span, [0, 1000], [500, 500])
    finally:
        enc.close()


def test_empty_file_inside_a_batch(oracle, hip):
    """a file of zero samples is a 43-byte header, alone (reference behaviour) and between other files of a batch"""
    p = S.make_params(2, 16, 48000, 16, 1, 8, 1, 1, 4096)
    a, b = W.music_like(2, 5000, 16, seed=1), W.music_like(2, 9000, 16, seed=2)
    empty = np.zeros((2, 0), np.int32)
    want_empty = oracle.encode_whole(p, empty)[1]
    assert len(want_empty) == 43
    enc = make_encoder(hip, p)
    try:
        assert enc.encode_whole(empty) == want_empty
        for files in ([a, empty, b], [empty, a], [a, empty], [empty], [empty, empty, b]):
            got = enc.encode_batch(files)
            for x, (rc, data) in zip(files, got):
                assert rc == 0 and data == oracle.encode_whole(p, x)[1]
    finally:
        enc.close()


def test_c4_at_its_real_batch_size(oracle, hip):
    """BASELINE config 4 as bench.py runs it on one GPU: 125 ten-second 48 kHz 16-bit stereo clips in ONE call, every clip
    byte-identical to the oracle's encode of that clip alone (the other batch tests use <= 10
    clips).  A few clips carry silence, a few another offset_lshift, so the pass splits; the oracle needs ~13 s for all."""
    p = S.make_params(2, 16, 48000, 16, 1, 8, 1, 1, 4096, cap=(2, 4096, 16, 1, 8))
    n = 480000
    pcms = []
    for i in range(125):
        pcm = S.synth_pcm(2, n, 16, 48000, seed=1000 + i) if i % 3 else W.music_like(2, n, 16, seed=2000 + i)
        if i % 31 == 7:
            pcm[:, 100000:160000] = 0                     # silence inside
        if i % 41 == 11:
            pcm = (pcm >> 18) << 18                       # offset_lshift 2
        if i == 124:
            pcm = pcm[:, :477777]                         # a ragged last clip
        pcms.append(np.ascontiguousarray(pcm))
    check_batch(oracle, hip, p, pcms, also_single=False)


# ------------------------------------------------------------------ batches on device-written block tables (round 4)

def _device_batch(hip, p, pcms, **options):
    """sla_hip_analyze_batch_device on the files laid out back to back: (trace, last_expand, starts)"""
    import torch
    nch = pcms[0].shape[0]
    starts, at = [], 0
    for x in pcms:
        starts.append(at)
        at += (x.shape[1] + 1023) // 1024 * 1024
    span = max(at, 1024)
    d_pcm = torch.zeros((nch, span), dtype=torch.int32, device="cuda")
    for s0, x in zip(starts, pcms):
        d_pcm[:, s0:s0 + x.shape[1]] = torch.from_numpy(x).cuda()
    torch.cuda.synchronize()
    enc = make_encoder(hip, p)
    for k, v in options.items():
        enc.set_option(k, v)
    return enc, d_pcm, span, starts


def _check_trace(oracle, p, tr, starts, pcms):
    b = 0
    for s0, x in zip(starts, pcms):
        ret, _, to = oracle.encode_trace(p, x)
        assert ret == 0
        k = to.num_blocks
        assert np.array_equal(tr.blk_start[b:b + k], to.blk_start[:k] + s0)
        assert np.array_equal(tr.blk_nsmpl[b:b + k], to.blk_nsmpl[:k]) and np.array_equal(tr.blk_type[b:b + k], to.blk_type[:k])
        comp = to.blk_type[:k] == 0
        for f in ("code", "kint", "rshift", "pitch", "rice_init"):
            assert np.array_equal(getattr(tr, f)[b:b + k][comp], getattr(to, f)[:k][comp]), f
        for j in np.nonzero(comp)[0]:
            a, n = int(to.blk_start[j]), int(to.blk_nsmpl[j])
            assert np.array_equal(tr.res_final[:, s0 + a:s0 + a + n], to.res_final[:, a:a + n])
        b += k
    assert b == tr.num_blocks


def test_batch_without_silence_takes_device_tables(oracle, hip):
    """no all-zero mask word and no silent file tail anywhere (k_batch_scan): the mask stays on the device, the block tables
    are written by k_expand (last_expand says so), the search tables are kept for the next batch of the same layout --
    and every file is the oracle's, also when the next batch has the same layout and other contents, and on host tables"""
    p = S.make_params(2, 16, 48000, 16, 1, 8, 1, 1, 4096, cap=(2, 4096, 16, 1, 8))
    lens = [30000, 4097, 12288, 2049, 20000, 4096 * 3 + 126]
    first = [W.music_like(2, n, 16, seed=170 + i) for i, n in enumerate(lens)]
    second = [S.synth_pcm(2, n, 16, 48000, seed=90 + i) for i, n in enumerate(lens)]
    enc, d_pcm, span, starts = _device_batch(hip, p, first)
    import torch
    try:
        enc.num_channels, enc.order, enc.ltm_order = 2, 16, 1
        hits0 = enc.last_expand()[2]
        enc.analyze_batch_device(d_pcm.data_ptr(), span, span, starts, lens)
        ex = enc.last_expand()
        assert ex[0] == ex[1] >= 1, ex                      # every pipeline chunk was launched from device-written tables
        _check_trace(oracle, p, enc.trace(), starts, first)
        for s0, x in zip(starts, second):
            d_pcm[:, s0:s0 + x.shape[1]] = torch.from_numpy(x).cuda()
        torch.cuda.synchronize()
        enc.analyze_batch_device(d_pcm.data_ptr(), span, span, starts, lens)
        ex2 = enc.last_expand()
        assert ex2[0] == ex2[1] >= 1 and ex2[2] == hits0 + 1, ex2      # same layout: the kept search tables served it
        _check_trace(oracle, p, enc.trace(), starts, second)
        # another layout: rebuilt, still right
        enc.analyze_batch_device(d_pcm.data_ptr(), span, span, starts[:3], lens[:3])
        _check_trace(oracle, p, enc.trace(), starts[:3], second[:3])
        enc.set_option("device_expand", 0)                   # host tables: same results
        enc.analyze_batch_device(d_pcm.data_ptr(), span, span, starts, lens)
        assert enc.last_expand()[0] == 0
        _check_trace(oracle, p, enc.trace(), starts, second)
    finally:
        enc.close()


def test_batch_with_a_silent_tail_or_a_zero_word_takes_the_mask(oracle, hip):
    """the two things k_batch_scan looks for: a file whose last super-frame (< 127 samples) is all zero -- a SILENT block no
    all-zero mask word betrays -- and an all-zero 64-sample word inside one file; either brings the batch's mask to the host
    (the super-frame hop needs it) and makes k_expand read the device's copy; option expand_silence = 0: host tables; bytes as
    the oracle's"""
    p = S.make_params(1, 16, 48000, 8, 1, 4, 0, 1, 4096)
    a = W.music_like(1, 4096 * 2 + 100, 16, seed=5)
    a[:, 4096 * 2:] = 0                                     # 100 zero samples = the whole last super-frame
    b = W.music_like(1, 9000, 16, seed=6)
    c = W.music_like(1, 4096 + 126, 16, seed=7)
    c[:, 4096:] = 0
    d = W.music_like(1, 4096 + 126, 16, seed=8)
    d[:, 4096:-1] = 0                                       # ... all but the last sample: not silent
    check_batch(oracle, hip, p, [a, b, c, d], also_single=False)
    e = W.music_like(1, 30000, 16, seed=9)
    e[:, 10048:10048 + 64] = 0                              # one aligned all-zero word (too short to be a block)
    check_batch(oracle, hip, p, [b, e, d], also_single=False)
    enc, d_pcm, span, starts = _device_batch(hip, p, [b, e, d])
    try:
        enc.num_channels, enc.order, enc.ltm_order = 1, 8, 1
        enc.analyze_batch_device(d_pcm.data_ptr(), span, span, starts, [x.shape[1] for x in (b, e, d)])
        assert enc.last_expand()[0] >= 1                     # round 4: device tables, k_expand reads the mask
        _check_trace(oracle, p, enc.trace(), starts, [b, e, d])
        enc.set_option("expand_silence", 0)
        enc.analyze_batch_device(d_pcm.data_ptr(), span, span, starts, [x.shape[1] for x in (b, e, d)])
        assert enc.last_expand()[0] == 0                     # host tables (rounds 2-3)
        _check_trace(oracle, p, enc.trace(), starts, [b, e, d])
    finally:
        enc.close()
    # (d's 125 zeros hold an aligned all-zero word: the mask route too.)  A tail of 100 samples that is NOT silent, no zero word:
    f = W.music_like(1, 4096 * 2 + 100, 16, seed=11)
    f[0, f[0] == 0] = 1 << 16
    g = b.copy()
    g[0, g[0] == 0] = 1 << 16
    enc, d_pcm, span, starts = _device_batch(hip, p, [g, f])
    try:
        enc.num_channels, enc.order, enc.ltm_order = 1, 8, 1
        enc.analyze_batch_device(d_pcm.data_ptr(), span, span, starts, [g.shape[1], f.shape[1]])
        assert enc.last_expand()[0] >= 1                     # nothing silent: device tables
        _check_trace(oracle, p, enc.trace(), starts, [g, f])
    finally:
        enc.close()


def test_big_batch_on_lanes(oracle, hip):
    """a batch big enough for the worker lanes (round 4: groups of consecutive files dealt out to handles of their own, uploads
    taking turns, everything behind them overlapping): every file's bytes are the one-piece batch's and the oracle's -- files
    of different lengths, one with silence, one whose low bits give another offset_lshift, one empty, one too small a buffer"""
    p = S.make_params(2, 16, 48000, 16, 1, 8, 1, 1, 4096, cap=(2, 4096, 16, 1, 8))
    rng = np.random.default_rng(77)
    lens = [int(x) for x in rng.integers(500000, 800000, size=18)]
    lens[5] = 4097
    lens[11] = 0
    pcms = [S.synth_pcm(2, max(n, 1), 16, 48000, seed=300 + i) if i % 3 else W.music_like(2, max(n, 1), 16, seed=300 + i) for i, n in enumerate(lens)]
    pcms[11] = np.zeros((2, 0), np.int32)
    pcms[3][:, 100000:160000] = 0
    pcms[7] = (pcms[7] >> 18) << 18                           # two more zero bits at the bottom: offset_lshift 2
    assert sum(lens) * 2 >= (16 << 20)
    caps = [8 * 2 * n + 65536 for n in lens]
    caps[9] = 1000                                            # too small: that file fails, the others do not
    enc = make_encoder(hip, p)
    one = make_encoder(hip, p)
    try:
        one.set_option("batch_lanes", 1)
        want = one.encode_batch(pcms, capacities=caps)
        for lanes in (4, 2, 6):
            enc.set_option("batch_lanes", lanes)
            got = enc.encode_batch(pcms, capacities=caps)
            assert [rc for rc, _ in got] == [rc for rc, _ in want], lanes
            assert [d for _, d in got] == [d for _, d in want], lanes
        assert want[9][0] != 0 and all(rc == 0 for i, (rc, _) in enumerate(want) if i != 9)
        for i in (0, 3, 5, 7, 11, 17):
            ret, ref = oracle.encode_whole(p, pcms[i])
            assert ret == 0 and got[i][1] == ref, i
        # the handle goes on: a small batch (plain path) and a single file
        again = enc.encode_batch(pcms[4:7])
        assert [d for _, d in again] == [d for _, d in want[4:7]]
        assert enc.encode_whole(pcms[5]) == want[5][1]
    finally:
        enc.close()
        one.close()

