"""world_size-2 gloo test (CPU) of the multi-GPU plumbing in sla_amd/dist.py: contiguous sharding of
independent units and the single all-gather that re-assembles the residual stream.  The per-rank
"device output" is produced by the oracle here (no GPU in this test); on the GPU box bench.py feeds the
same helpers with the HIP path's residual planes over RCCL."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import slalibs as S
from sla_amd import dist as sdist


def test_shard_units_partition():
    for n in (0, 1, 7, 8, 9, 1000, 7032):
        for world in (1, 2, 3, 8):
            spans = [sdist.shard_units(n, world, r) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            sizes = [hi - lo for lo, hi in spans]
            assert max(sizes) - min(sizes) <= 1


def _worker(rank, world, port, clips, n, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        o = S.oracle()
        p = S.make_params(2, 16, 48000, 16, 1, 8, 1, 1, 4096, cap=(2, 4096, 16, 1, 8))
        lo, hi = sdist.shard_units(clips, world, rank)
        mine = []
        for c in range(lo, hi):                        # this rank's shard of independent clips
            pcm = S.synth_pcm(2, n, 16, 48000, seed=1000 + c)
            ret, data, tr = o.encode_trace(p, pcm)
            assert ret == 0
            mine.append(tr.res_final.copy())
        per_rank = (clips + world - 1) // world
        planes = np.zeros((per_rank, 2, n), np.int32)  # padded to equal size for the collective
        planes[:len(mine)] = np.stack(mine) if mine else planes[:0]
        gathered = sdist.all_gather_planes(torch.from_numpy(planes.reshape(per_rank * 2, n)))
        # the overlapped form bench.py uses at N > 1: two plane sets in flight, settled before reuse
        outs, works = [], []
        for k in range(2):
            o2, w2 = sdist.all_gather_planes(torch.from_numpy(planes.reshape(per_rank * 2, n)) + k, async_op=True)
            outs.append(o2); works.append(w2)
        for k in range(2):
            works[k].wait()
            assert torch.equal(outs[k], gathered + k)
        t = sdist.max_over_ranks(float(rank + 1), "cpu")
        assert t == float(world)
        q.put((rank, gathered.numpy().reshape(world, per_rank, 2, n).copy()))
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_two_rank_gather_reassembles_residual_stream():
    world, clips, n = 2, 5, 9000
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, clips, n, q)) for r in range(world)]
    for pr in procs:
        pr.start()
    results = dict(q.get(timeout=120) for _ in range(world))
    for pr in procs:
        pr.join(60)
        assert pr.exitcode == 0
    o = S.oracle()
    p = S.make_params(2, 16, 48000, 16, 1, 8, 1, 1, 4096, cap=(2, 4096, 16, 1, 8))
    assert np.array_equal(results[0], results[1])      # every rank holds the whole stream
    g = results[0]
    for c in range(clips):
        owner = 0 if c < 3 else 1
        slot = c - (0 if owner == 0 else 3)
        pcm = S.synth_pcm(2, n, 16, 48000, seed=1000 + c)
        _, _, tr = o.encode_trace(p, pcm)
        assert np.array_equal(g[owner, slot], tr.res_final)


# ------------------------------------------------------------------ ONE file over several ranks (sla_amd/dist.encode_sharded)

class OracleShardBackend:
    """stands in for the GPU in the CPU tests: the oracle scans and encodes the ranges; everything between -- piece
    cuts, exchanges, sla_hip_shard_bounds, sla_hip_shard_header, assembly -- is the product's code"""

    def __init__(self, oracle, p, pcm):
        self.o, self.p, self.pcm = oracle, p, pcm

    def scan(self, lo, hi):
        x = self.pcm[:, lo:hi]
        orw = int(np.bitwise_or.reduce(x.view(np.uint32).ravel())) if x.size else 0
        sh = 32 - self.p.bits_per_sample
        v = x >> sh
        if self.p.ch_process_method == 1:
            nz = (((v[0].astype(np.int64) + v[1]) >> 1) != 0) | (v[0] != v[1])
        else:
            nz = (v != 0).any(axis=0)
        bits = np.zeros(((hi - lo + 63) // 64) * 64, np.uint8)
        bits[:hi - lo] = nz
        return orw, np.packbits(bits.reshape(-1, 64), axis=1, bitorder="little").view(np.uint64).ravel()

    def encode_range(self, lo, hi, file_or):
        if hi <= lo:
            return b""
        ntz = (file_or & -file_or).bit_length() - 1 if file_or else 32
        lshift = self.p.bits_per_sample - (32 - ntz) if file_or else 0
        ret, data = self.o.encode_range(self.p, np.ascontiguousarray(self.pcm[:, lo:hi]), lshift)
        assert ret == 0
        return data


def sharded_cases():
    """files whose sharding has something to get wrong: leading silence (shifts every later super-frame), silence
    across a piece cut, a quiet piece whose own OR word would give another offset_lshift, a silent tail"""
    out = []
    a = S.synth_pcm(1, 70000, 16, 48000, seed=5)
    a[:, :3000] = 0
    a[:, 33000:37500] = 0
    a[:, 66000:] = 0
    out.append(("mono16", a, S.make_params(1, 16, 48000, 16, 1, 8, 0, 1, 4096, cap=(1, 4096, 16, 1, 8))))
    b = S.synth_pcm(2, 90000, 24, 48000, seed=6)
    b[:, :45000] &= ~np.int32(0xFFF)                  # the first half alone would shift by 4 more bits
    b[:, 20000:23000] = 0
    b[:, 44000:47000] = 0
    out.append(("stereo24ms", b, S.make_params(2, 24, 48000, 32, 3, 8, 1, 1, 8192, cap=(2, 8192, 32, 3, 8))))
    # BASELINE config 5's shape (8 channels, 24 bit, 96 kHz, order 48, 8192-sample blocks), the file's own length split over the
    # ranks = bench.py --scaling strong: a cut that falls inside a super-frame, a silent stretch on one side of it
    c = S.synth_pcm(8, 41000, 24, 96000, seed=7)
    c[:, 26000:29000] = 0
    out.append(("c5shape8ch", c, S.make_params(8, 24, 96000, 48, 3, 8, 0, 1, 8192, cap=(8, 8192, 48, 3, 8))))
    return out


def _shard_worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        o = S.oracle()
        res = []
        for name, pcm, p in sharded_cases():
            got = sdist.encode_sharded(OracleShardBackend(o, p, pcm), pcm.shape[1], p.max_block_samples)
            res.append((name, got))
        q.put((rank, res))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_one_file_over_ranks_is_the_single_rank_file(world):
    """world-size 2 and 3 over gloo: the assembled .sla of a file with leading and interior silence equals the
    oracle's encode of the whole file, byte for byte"""
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_shard_worker, args=(r, world, port, q)) for r in range(world)]
    for pr in procs:
        pr.start()
    got = dict(q.get(timeout=600) for _ in range(world))
    for pr in procs:
        pr.join(timeout=60)
        assert pr.exitcode == 0
    o = S.oracle()
    for (name, pcm, p), (gname, data) in zip(sharded_cases(), got[0]):
        ret, want = o.encode_whole(p, pcm)
        assert ret == 0 and gname == name
        assert data == want, name
    assert all(d is None for r in range(1, world) for _, d in got[r])


def test_shard_bounds_follow_the_silence_hop():
    """sla_hip_shard_bounds (host arithmetic of the product, no GPU): bounds are super-frame starts of the reference's hop"""
    import sla_amd
    n, maxb = 100000, 4096
    nz = np.ones(n, np.uint8)
    nz[:3000] = 0                      # leading silence: one SILENT block of 3000, everything behind it shifted
    nz[50000:50100] = 0                # too short to matter
    nz[70000:80000] = 0                # a long run: super-frames inside it are silent blocks of up to 4096
    bits = np.zeros((n + 63) // 64 * 64, np.uint8); bits[:n] = nz
    mask = np.packbits(bits.reshape(-1, 64), axis=1, bitorder="little").view(np.uint64).ravel()
    starts, pos = [], 0
    while pos < n:
        starts.append(pos)
        win, minb = min(maxb, n - pos), min(2048, n - pos)
        run = 0
        while run < win and not nz[pos + run]:
            run += 1
        pos += run if run >= minb else win
    for world in (1, 2, 3, 8, 64):
        b = sla_amd.shard_bounds(n, maxb, mask, world)
        assert b[0] == 0 and b[-1] == n and all(x <= y for x, y in zip(b, b[1:]))
        for r in range(1, world):
            target = (n * r + world - 1) // world
            assert b[r] == min([s for s in starts if s >= target] + [n])


def test_shard_bounds_without_a_mask_when_nothing_is_silent():
    """sla_hip_shard_scan_counts path: no rank counted an all-zero mask word -> sla_hip_shard_bounds(NULL) = the hop of a
    file without silence = what the full mask gives when it has no silence run of a minimum block"""
    import sla_amd
    for n, maxb in ((100000, 4096), (5000, 4096), (4096 * 7, 4096), (123457, 16384), (3, 2048)):
        bits = np.ones((n + 63) // 64 * 64, np.uint8)
        bits[n:] = 0
        # short zero runs (below a mask word, and a whole word or two but below the 2048-sample minimum block)
        if n > 70000:
            bits[50000:50060] = 0
            bits[64 * 1000:64 * 1002] = 0
        mask = np.packbits(bits.reshape(-1, 64), axis=1, bitorder="little").view(np.uint64).ravel()
        for world in (1, 2, 3, 8):
            assert sla_amd.shard_bounds(n, maxb, None, world) == sla_amd.shard_bounds(n, maxb, mask, world)


def test_upload_range_contains_every_rank_range():
    """the piece a rank uploads must contain the range sla_hip_shard_bounds later gives it, also when the
    block size is not a multiple of 1024 (bounds[r+1] sits up to one block behind the UNFLOORED target) and when a
    silence run has moved the super-frame starts off the block grid"""
    import sla_amd
    rng = np.random.default_rng(5)
    cases = [(12002, 2, 3000), (250001, 5, 5000), (100000, 3, 2048), (1 << 20, 8, 4096), (77777, 7, 16384), (9000, 8, 4096)]
    cases += [(int(rng.integers(2048, 400000)), int(rng.integers(1, 9)), int(rng.integers(2048, 16385))) for _ in range(200)]
    for n, world, maxb in cases:
        for with_silence in (False, True):
            mask = None
            if with_silence:
                bits = np.ones(n + 64, bool)
                a = int(rng.integers(0, max(n - 3000, 1)))
                bits[a:a + int(rng.integers(2048, 3000))] = False      # one silence run of at least a minimum block
                mask = np.packbits(bits[:(n + 63) // 64 * 64].reshape(-1, 64)[:, ::-1], axis=1).view(">u8").astype(np.uint64).ravel()
            bounds = sla_amd.shard_bounds(n, maxb, mask, world)
            assert bounds[0] == 0 and bounds[-1] == n
            for r in range(world):
                lo, top = sdist.upload_range(n, world, r, maxb)
                assert lo <= bounds[r] and bounds[r + 1] <= top, (n, world, maxb, r, bounds, lo, top)
                assert sdist.scan_piece(n, world, r)[0] == lo


@pytest.mark.timeout(300)
def test_bench_launches_its_own_ranks():
    """`python bench.py --gpus 2` without WORLD_SIZE starts the two ranks itself (torch.distributed.run as a child
    process) and relays rank 0's line; a rank whose WORLD_SIZE differs from --gpus exits non-zero"""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--backend", "gloo", "--launch-check",
                        "--scaling", "strong"], capture_output=True, text=True, env=env, timeout=240)
    assert r.returncode == 0, r.stderr[-2000:]
    line = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    assert line["n_gpus"] == 2 and line["ranks_seen"] == 2 and line["scaling"] == "strong"
    env["WORLD_SIZE"], env["RANK"] = "1", "0"
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--launch-check"],
                       capture_output=True, text=True, env=env, timeout=120)
    assert r.returncode != 0 and "WORLD_SIZE" in r.stderr
