"""Multi-GPU plumbing of the encode path (one process per GPU, torch.distributed; backend "nccl" is
RCCL over xGMI on ROCm, "gloo" is used by the CPU tests).

Units of work are independent (every SLA block resets all filter and coder state, reference
src/SLAEncoder.c:594-659), so ranks take contiguous shards of units with no data-path exchange
until the residual stream is re-assembled for the serial bit-pack: ONE all-gather per step."""
import torch
import torch.distributed as dist


def shard_units(num_units, world, rank):
    """contiguous, balanced [lo, hi) of `num_units` for `rank` (first `num_units % world` ranks get one more)"""
    base, extra = divmod(num_units, world)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def all_gather_planes(planes, out=None, async_op=False):
    """planes: int32 tensor [C, stride] holding this rank's residual planes.  Returns [world, C, stride]
    with every rank's planes (one collective; RCCL ring over xGMI when the backend is nccl).
    async_op=True returns (out, work): the collective travels while the caller analyses its next shard
    into OTHER planes; work.wait() before `out` is read or `planes` are written again."""
    world = dist.get_world_size()
    shape = tuple(planes.shape)
    if out is None:
        out = torch.empty((world,) + shape, dtype=planes.dtype, device=planes.device)
    # concatenated form (world*C, stride): accepted by both RCCL and gloo
    work = dist.all_gather_into_tensor(out.view((world * shape[0],) + shape[1:]), planes.contiguous(), async_op=async_op)
    return (out, work) if async_op else out


def max_over_ranks(seconds, device):
    t = torch.tensor([seconds], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())
