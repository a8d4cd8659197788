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
