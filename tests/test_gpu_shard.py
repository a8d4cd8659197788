"""GPU side of the one-file-over-several-GPUs path (include/sla_hip.h "one file, several GPUs", sla_amd/dist.py):
the ranks are played one after the other on the single GPU of the test box, each with its own encoder handle, through
exactly the calls a rank makes (sla_hip_shard_scan / _bounds / _analyze, device pack, sla_hip_shard_header).  The
assembled file must be the oracle's -- and the single-GPU SLAEncoder_EncodeWhole's -- byte for byte."""
import numpy as np
import pytest

import slalibs as S
from test_dist_gloo import sharded_cases

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def hip():
    import torch
    if not torch.cuda.is_available():
        pytest.skip("needs an MI355X (the HIP path has no CPU fallback)")
    import sla_amd
    return sla_amd


@pytest.fixture(scope="module")
def oracle():
    return S.oracle()


def _encoder(hip, p):
    enc = hip.Encoder(p.cap_channels, p.cap_block_samples, p.cap_parcor_order, p.cap_longterm_order, p.cap_lms_order)
    enc.set_wave_format(p.num_channels, p.bits_per_sample, p.sampling_rate)
    enc.set_encode_parameter(p.parcor_order, p.longterm_order, p.lms_order, p.ch_process_method, p.window_type, p.max_block_samples)
    return enc


@pytest.mark.parametrize("world", [1, 2, 3, 5, 8])
@pytest.mark.parametrize("case", [0, 1])
def test_sharded_file_equals_single_gpu_file(oracle, hip, world, case):
    from sla_amd import dist as sdist
    name, pcm, p = sharded_cases()[case]
    ret, want = oracle.encode_whole(p, pcm)
    assert ret == 0
    encs = [_encoder(hip, p) for _ in range(world)]
    try:
        backs = [sdist.HipShardBackend(e, pcm, p.max_block_samples) for e in encs]
        got, bounds = sdist.encode_sharded_serial(backs, pcm.shape[1], p.max_block_samples)
        assert got == want, (name, world, bounds)
        whole = encs[0].encode_whole(pcm)
        assert whole == want
    finally:
        for e in encs:
            e.close()


def test_more_ranks_than_superframes_and_silent_file(oracle, hip):
    """a file of two super-frames over 8 ranks (six ranks own nothing), and an all-zero file (OR word 0)"""
    from sla_amd import dist as sdist
    p = S.make_params(1, 16, 48000, 16, 1, 8, 0, 1, 4096, cap=(1, 4096, 16, 1, 8))
    for pcm in (S.synth_pcm(1, 6000, 16, 48000, seed=9), np.zeros((1, 9000), np.int32)):
        ret, want = oracle.encode_whole(p, pcm)
        assert ret == 0
        encs = [_encoder(hip, p) for _ in range(8)]
        try:
            got, bounds = sdist.encode_sharded_serial([sdist.HipShardBackend(e, pcm, 4096) for e in encs], pcm.shape[1], 4096)
            assert got == want, bounds
        finally:
            for e in encs:
                e.close()


def test_range_with_foreign_or_word_is_refused(hip):
    """sla_hip_shard_analyze checks that the range can be part of a file with the given OR word"""
    import torch
    p = S.make_params(1, 16, 48000, 16, 1, 8, 0, 1, 4096, cap=(1, 4096, 16, 1, 8))
    pcm = S.synth_pcm(1, 8192, 16, 48000, seed=3)
    enc = _encoder(hip, p)
    try:
        d = torch.from_numpy(pcm).cuda()
        with pytest.raises(hip.SlaError):
            enc.shard_analyze(d.data_ptr(), 8192, 8192, 0x00F00000)       # the samples have bits outside that word
    finally:
        enc.close()


def test_scan_counts_decide_whether_the_mask_is_needed(oracle, hip):
    """sla_hip_shard_scan_counts: OR word + number of all-zero mask words of a piece; a file without such words takes the
    12-byte exchange (encode_sharded_serial does, through HipShardBackend.scan_counts), one with a silence run the mask"""
    import torch
    from sla_amd import dist as sdist
    p = S.make_params(2, 16, 48000, 16, 1, 8, 1, 1, 4096, cap=(2, 4096, 16, 1, 8))
    loud = S.synth_pcm(2, 60000, 16, 48000, seed=3)
    loud[loud == 0] = 1 << 16
    gap = loud.copy()
    gap[:, 20000:26000] = 0
    for pcm, silent in ((loud, False), (gap, True)):
        pcm = np.ascontiguousarray(pcm)
        enc = _encoder(hip, p)
        try:
            n = pcm.shape[1]
            stride = (n + 63) // 64 * 64
            d = torch.zeros((2, stride), dtype=torch.int32, device="cuda")
            d[:, :n] = torch.from_numpy(pcm).cuda()
            torch.cuda.synchronize()
            orw, zeros = enc.shard_scan_counts(d.data_ptr(), stride, n)
            orw2, mask = enc.shard_scan(d.data_ptr(), stride, n)
            assert orw == orw2 and zeros == int((mask[:n // 64] == 0).sum()) + (1 if n % 64 and mask[n // 64] == 0 else 0)
            assert (zeros != 0) == silent
        finally:
            enc.close()
        ret, want = oracle.encode_whole(p, pcm)
        assert ret == 0
        for world in (2, 3):
            encs = [_encoder(hip, p) for _ in range(world)]
            try:
                got, _ = sdist.encode_sharded_serial([sdist.HipShardBackend(e, pcm, 4096) for e in encs], pcm.shape[1], 4096)
                assert got == want
            finally:
                for e in encs:
                    e.close()


@pytest.mark.parametrize("maxb,n,world", [(3000, 12002, 2), (5000, 250001, 5), (2048 + 64, 40000, 3), (4096, 30000, 3)])
def test_block_sizes_off_the_1024_grid_and_dealigned_hops(oracle, hip, maxb, n, world):
    """bounds[r+1] may sit up to one block behind the unfloored target -- 1022 samples past what the ranks
    used to upload when the block size is not a multiple of 1024, or when a silence run moves the super-frame starts
    off the grid; the third file ends in a short all-zero tail (a SILENT block the no-silence shortcut must not miss)"""
    from sla_amd import dist as sdist
    p = S.make_params(1, 16, 48000, 8, 1, 8, 0, 1, maxb, cap=(1, 16384, 16, 1, 8))
    files = [S.synth_pcm(1, n, 16, 48000, seed=world)]
    quiet = S.synth_pcm(1, n, 16, 48000, seed=world + 1)
    quiet[:, 5000:5000 + 2500] = 0                       # a silence run of 2500 samples: every later super-frame start moves
    files.append(quiet)
    tail = S.synth_pcm(1, n - n % maxb + 60, 16, 48000, seed=world + 2)
    tail[:, -60:] = 0                                   # last super-frame: 60 zero samples = a SILENT block without any all-zero mask word
    files.append(tail)
    for pcm in files:
        ret, want = oracle.encode_whole(p, pcm)
        assert ret == 0
        encs = [_encoder(hip, p) for _ in range(world)]
        try:
            backs = [sdist.HipShardBackend(e, pcm, maxb) for e in encs]
            got, bounds = sdist.encode_sharded_serial(backs, pcm.shape[1], maxb)
            assert got == want, (maxb, n, world, bounds)
        finally:
            for e in encs:
                e.close()
