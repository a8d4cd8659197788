"""Block tables written on the device (k_expand, option "device_expand"): the block stage is launched from two counts,
the host's own tables follow under the kernels.  Same bytes as the host-table route and as the oracle; the route falls
back to host tables chunk-wise when the device cannot certify a partition.  Input with all-zero mask words (where a block
inside a super-frame can be SILENT) takes it too since round 4: k_expand reads the prepass mask (option "expand_silence").

Reference: the walk over the super-frames that numbers the blocks, src/SLAEncoder.c:846-869."""
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


def _encode(hip, p, pcm, **options):
    enc = hip.Encoder(p.cap_channels, p.cap_block_samples, p.cap_parcor_order, p.cap_longterm_order, p.cap_lms_order)
    try:
        for k, v in options.items():
            enc.set_option(k, v)
        enc.set_wave_format(p.num_channels, p.bits_per_sample, p.sampling_rate)
        enc.set_encode_parameter(p.parcor_order, p.longterm_order, p.lms_order, p.ch_process_method,
                                 p.window_type, p.max_block_samples)
        data = enc.encode_whole(pcm)
        return data, enc.last_expand()[:2], enc.last_counters()
    finally:
        enc.close()


CASES = [
    # nch, bits, rate, order, ms, maxb, samples, kind
    (1, 16, 48000, 16, 0, 4096, 48000 * 60, "bench"),          # the C2 shape: two chunks
    (2, 24, 48000, 32, 1, 4096, 48000 * 40 + 1234, "bench"),   # ragged last super-frame
    (2, 16, 48000, 16, 1, 4096, 480000, "music"),              # a clip: one chunk
    (3, 24, 96000, 48, 0, 8192, 96000 * 12 + 77, "music"),     # a last super-frame of 77 samples
    (1, 24, 44100, 8, 0, 2048, 44100 * 30, "noise"),
    (2, 20, 48000, 24, 1, 16384, 48000 * 30 + 5000, "music"),
]


def _signal(kind, nch, n, bits, seed):
    if kind == "bench":
        return S.synth_pcm(nch, n, bits, 48000, seed=seed)
    if kind == "music":
        return W.music_like(nch, n, bits, seed=seed)
    rng = np.random.default_rng(seed)
    x = rng.integers(-(1 << (bits - 3)), 1 << (bits - 3), size=(nch, n), dtype=np.int64)
    return (x << (32 - bits)).astype(np.int32)


@pytest.mark.parametrize("nch,bits,rate,order,ms,maxb,n,kind", CASES)
def test_device_tables_give_the_host_tables_bytes(oracle, hip, nch, bits, rate, order, ms, maxb, n, kind):
    pcm = _signal(kind, nch, n, bits, seed=order + nch)
    p = S.make_params(nch, bits, rate, order, 1, 8, ms, 1, maxb)
    host, eh, _ = _encode(hip, p, pcm, device_expand=0, stream=0)
    dev, ed, cnt = _encode(hip, p, pcm, device_expand=1, stream=0)
    assert eh[0] == 0
    assert dev == host
    # the certified block kernels queued with the searches (short files) or after the counts; everything on one stream
    assert _encode(hip, p, pcm, device_expand=1, stream=0, prelaunch=0)[0] == host
    assert _encode(hip, p, pcm, device_expand=1, stream=0, one_stream=1)[0] == host
    # every chunk of a file without silence whose partitions all certify runs from device tables
    if cnt[1] == 0:
        assert ed[0] == ed[1] and ed[1] >= 1, ed
    if n <= 48000 * 40 + 1234:
        ret, want = oracle.encode_whole(p, pcm)
        assert ret == 0 and dev == want


@pytest.mark.parametrize("chunks", [1, 2, 3, 5])
def test_chunk_counts(hip, chunks):
    pcm = S.synth_pcm(2, 48000 * 90, 16, 48000, seed=7)
    p = S.make_params(2, 16, 48000, 16, 1, 8, 1, 1, 4096)
    host, _, _ = _encode(hip, p, pcm, device_expand=0, stream=0, chunks=chunks)
    dev, ed, _ = _encode(hip, p, pcm, device_expand=1, stream=0, chunks=chunks)
    assert dev == host
    assert ed == (chunks, chunks)


def test_uncertified_partitions_take_the_host_tables(oracle, hip):
    """plan_margin = 1e30: k_plan certifies nothing, k_expand reports the chunk as not valid, the host route runs"""
    pcm = W.music_like(2, 300000, 16, seed=3)
    p = S.make_params(2, 16, 48000, 16, 1, 8, 1, 1, 4096)
    ret, want = oracle.encode_whole(p, pcm)
    assert ret == 0
    got, ed, cnt = _encode(hip, p, pcm, device_expand=1, plan_margin=1e30, stream=0, chunks=2)
    assert got == want
    assert ed[0] == 0 and cnt[1] > 0


def test_silence_takes_device_tables_too(oracle, hip):
    """all-zero mask words: a block inside a super-frame can be SILENT.  Round 4: k_expand reads the mask and gives such a
    block a number but no group; option expand_silence = 0 keeps the host tables of rounds 2-3"""
    pcm = W.music_like(2, 400000, 16, seed=5)
    pcm[:, 100000:140000] = 0
    pcm[:, 390000:] = 0
    p = S.make_params(2, 16, 48000, 16, 1, 8, 1, 1, 4096)
    ret, want = oracle.encode_whole(p, pcm)
    assert ret == 0
    got, ed, cnt = _encode(hip, p, pcm, device_expand=1, stream=0)
    assert got == want
    if cnt[1] == 0:
        assert ed[0] == ed[1] >= 1, ed
    got, ed, _ = _encode(hip, p, pcm, device_expand=1, expand_silence=0, stream=0)
    assert got == want
    assert ed[0] == 0


SILENCE_CASES = [
    # nch, bits, order, ms, maxb, super-frames
    (1, 16, 16, 0, 4096, 40),
    (2, 24, 32, 1, 16384, 24),
    (3, 20, 48, 0, 8192, 20),
    (2, 16, 8, 1, 2048, 60),
]


@pytest.mark.parametrize("nch,bits,order,ms,maxb,nsf", SILENCE_CASES)
def test_silent_blocks_inside_searched_superframes(oracle, hip, nch, bits, order, ms, maxb, nsf):
    """zero runs that do NOT begin where a super-frame begins: the super-frame is searched like any other and the blocks of
    its partition that are all zero become SILENT blocks (src/SLAEncoder.c:392-408) -- a block number, no group -- between
    compressed ones; zero runs that do begin there move the grid.  Device tables (k_expand with the mask), host tables and the
    oracle agree on bytes and on the block table"""
    rng = np.random.default_rng(nch * 1000 + maxb)
    n = maxb * nsf + int(rng.integers(1, maxb))
    pcm = W.music_like(nch, n, bits, seed=order + maxb)
    nz = (pcm != 0).any(axis=0)
    pcm[0, ~nz] = 1 << (32 - bits)                         # no accidental zero sample
    at = 0
    for k in range(nsf):
        lo = k * maxb
        kind = k % 6
        if kind == 1:                                      # the second half of the window
            pcm[:, lo + maxb // 2:lo + maxb] = 0
        elif kind == 2 and maxb >= 4096:                   # a middle piece on the search grid
            pcm[:, lo + 1024:lo + 1024 + 2048] = 0
        elif kind == 3:                                    # everything but the first sample
            pcm[:, lo + 1:lo + maxb] = 0
        elif kind == 4:                                    # off the grid: from somewhere to the end of the window and beyond
            s0 = lo + int(rng.integers(1, maxb // 2))
            pcm[:, s0:lo + maxb + int(rng.integers(0, 3000))] = 0
    p = S.make_params(nch, bits, 48000, order, 1, 8, ms, 1, maxb)
    ret, want = oracle.encode_whole(p, pcm)
    assert ret == 0
    enc = hip.Encoder(p.cap_channels, p.cap_block_samples, p.cap_parcor_order, p.cap_longterm_order, p.cap_lms_order)
    try:
        enc.set_option("stream", 0)
        enc.set_wave_format(nch, bits, 48000)
        enc.set_encode_parameter(order, 1, 8, ms, 1, maxb)
        got = enc.encode_whole(pcm)
        ed, cnt = enc.last_expand()[:2], enc.last_counters()
        tr = enc.trace(want_residuals=False)
    finally:
        enc.close()
    assert got == want
    if cnt[1] == 0:
        assert ed[0] == ed[1] >= 1, ed
    nb = tr.num_blocks
    typ, start, ln = tr.blk_type[:nb], tr.blk_start[:nb], tr.blk_nsmpl[:nb]
    # SILENT blocks that follow a compressed block of fewer than maxb samples: inside a searched super-frame (or right behind
    # a ragged one -- both kinds are wanted)
    inner = [(int(start[b]), int(ln[b])) for b in range(1, nb) if typ[b] == 1 and typ[b - 1] != 1 and ln[b - 1] < maxb]
    assert len(inner) >= 2 or maxb == 2048, (inner, nb)          # (maxb = the minimum block length: a super-frame is one block)
    assert _encode(hip, p, pcm, device_expand=0, stream=0)[0] == want
    assert _encode(hip, p, pcm, expand_silence=0, stream=0)[0] == want
    assert _encode(hip, p, pcm, chunks=3, stream=0)[0] == want
    assert _encode(hip, p, pcm, stream=0, prelaunch=0, one_stream=1)[0] == want


def test_silent_last_superframe(oracle, hip):
    """no all-zero mask word, but the file ends in a few zero samples that make a SILENT last super-frame"""
    n = 4096 * 30 + 40
    pcm = W.music_like(1, n, 16, seed=9)
    pcm[:, 4096 * 30:] = 0
    p = S.make_params(1, 16, 48000, 16, 1, 8, 0, 1, 4096)
    ret, want = oracle.encode_whole(p, pcm)
    assert ret == 0
    got, ed, _ = _encode(hip, p, pcm, device_expand=1, stream=0)
    assert got == want


def test_expand_launcher_rejects_bad_arguments(hip):
    L = hip.lib()
    assert L.sla_hip_launch_expand(None, 0, None, None, None, 1, 0, None, None, 0, None, None, None, None, None, 0, None, 1, None) != 0
    assert L.sla_hip_launch_expand_masked(None, 0, None, None, None, 1, 0, None, None, 0, None, None, None, None, None, 0, None, 1, None, None) != 0


def test_handle_reuse_across_routes(oracle, hip):
    """one handle, files of different kinds one after the other: the sequence word and the running counters restart"""
    p = S.make_params(2, 16, 48000, 16, 1, 8, 1, 1, 4096)
    enc = hip.Encoder(p.cap_channels, p.cap_block_samples, p.cap_parcor_order, p.cap_longterm_order, p.cap_lms_order)
    try:
        enc.set_option("stream", 0)
        enc.set_wave_format(2, 16, 48000)
        enc.set_encode_parameter(16, 1, 8, 1, 1, 4096)
        for i, n in enumerate([200000, 4096 * 300, 50001, 200000]):
            pcm = W.music_like(2, n, 16, seed=20 + i)
            if i == 2:
                pcm[:, 10000:30000] = 0
            ret, want = oracle.encode_whole(p, pcm)
            assert ret == 0
            assert enc.encode_whole(pcm) == want
            assert enc.last_expand()[0] >= 1                # (i == 2, silence: device tables through the mask since round 4)
    finally:
        enc.close()


def test_kept_search_tables(oracle, hip):
    """option "table_cache": the search tables of a file without silence serve the next file of the same length and
    parameters; any other file in between (another length, silence, other parameters, a batch) replaces or drops them"""
    p = S.make_params(2, 16, 48000, 16, 1, 8, 1, 1, 4096)
    enc = hip.Encoder(p.cap_channels, p.cap_block_samples, p.cap_parcor_order, p.cap_longterm_order, p.cap_lms_order)
    try:
        enc.set_option("stream", 0)
        enc.set_wave_format(2, 16, 48000)
        enc.set_encode_parameter(16, 1, 8, 1, 1, 4096)
        hits = []
        plan = [(200000, False), (200000, False), (200000, True), (200000, False), (200000, False), (150001, False), (200000, False),
                (200000, False)]
        for i, (n, gap) in enumerate(plan):
            pcm = W.music_like(2, n, 16, seed=40 + i)
            if gap:
                pcm[:, 50000:80000] = 0
            ret, want = oracle.encode_whole(p, pcm)
            assert ret == 0
            assert enc.encode_whole(pcm) == want
            hits.append(enc.last_expand()[2])
        # file 1 reuses file 0's tables; so does the file with silence at first (the tables are the guess made while the
        # prepass runs) -- and drops them when the prepass reports silence; 3 rebuilds, 4 reuses; another length replaces them
        assert hits == [0, 1, 2, 2, 3, 3, 3, 4], hits
        # the searches of files 1, 2, 4 and 7 went out on the guess "like the file before"; file 2 (silence) proved it wrong
        assert enc.last_expand()[3] == 1
        # same shape, another OR word (two low bits always zero: offset_lshift 2): the guess is wrong, the bytes are not
        pcm = (W.music_like(2, 200000, 16, seed=77) >> 18) << 18
        ret, want = oracle.encode_whole(p, pcm)
        assert ret == 0 and enc.encode_whole(pcm) == want
        assert enc.last_expand()[2:] == (5, 2)
        pcm = W.music_like(2, 200000, 16, seed=78)
        ret, want = oracle.encode_whole(p, pcm)
        assert ret == 0 and enc.encode_whole(pcm) == want
        assert enc.last_expand()[2:] == (6, 3)
        # other parameters: new tables, and the same bytes as a fresh handle
        enc.set_encode_parameter(8, 1, 8, 1, 1, 2048)
        p2 = S.make_params(2, 16, 48000, 8, 1, 8, 1, 1, 2048)
        pcm = W.music_like(2, 200000, 16, seed=99)
        ret, want = oracle.encode_whole(p2, pcm)
        assert ret == 0 and enc.encode_whole(pcm) == want
        assert enc.last_expand()[2] == 6
        enc.set_option("table_cache", 0)
        assert enc.encode_whole(pcm) == want and enc.last_expand()[2] == 6
    finally:
        enc.close()


def test_guess_meets_a_silent_last_superframe(oracle, hip):
    """two files of one shape on one handle; the second one ends in a few zero samples that make its last super-frame
    SILENT (no all-zero mask word reports that: the tail words of the mask do) -- the searches launched on the guess
    "like the file before" are thrown away, the tables rebuilt, the bytes are the oracle's; then the first file again"""
    n = 4096 * 48 + 40
    p = S.make_params(2, 16, 48000, 16, 1, 8, 1, 1, 4096)
    a = W.music_like(2, n, 16, seed=61)
    b = W.music_like(2, n, 16, seed=62)
    b[:, 4096 * 48:] = 0
    enc = hip.Encoder(p.cap_channels, p.cap_block_samples, p.cap_parcor_order, p.cap_longterm_order, p.cap_lms_order)
    try:
        enc.set_option("stream", 0)
        enc.set_wave_format(2, 16, 48000)
        enc.set_encode_parameter(16, 1, 8, 1, 1, 4096)
        for pcm in (a, a, b, a, a, b, b):
            ret, want = oracle.encode_whole(p, pcm)
            assert ret == 0
            assert enc.encode_whole(pcm) == want
        assert enc.last_expand()[3] >= 2          # both first meetings with b were wrong guesses
    finally:
        enc.close()
