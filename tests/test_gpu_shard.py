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
