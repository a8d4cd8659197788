"""One file over the GPUs of a node: one process per GPU, torch.distributed (backend "nccl" is RCCL over xGMI on
ROCm; "gloo" is what the CPU tests use).

Blocks are independent (every SLA block resets all filter and coder state, reference src/SLAEncoder.c:594-659), so
the super-frames of ONE file shard over the ranks.  Three facts of the whole file are agreed on first -- include/
sla_hip.h, "one file, several GPUs", is the contract this module drives:

  1. scan        rank r scans its piece of the file            -> OR word (offset_lshift, src/SLAEncoder.c:425-455)
                                                                  + 1 bit per sample "not silent" (:392-408)
  2. exchange    the OR of the ranks' words (4 bytes each, all-gathered: RCCL has no bitwise reduction) and all-gather of
                 the mask pieces (N/8 bytes in total)
  3. bounds      every rank runs the same host arithmetic (sla_hip_shard_bounds): rank r owns super-frames
                 [bounds[r], bounds[r+1]) of the whole file's hop over silence runs (:846-869)
  4. encode      the hot path on the rank's own range with the FILE's OR word, then the device bit-pack
  5. assemble    all-gather of the compressed images (sizes first); rank 0 joins them under one header (:920-926)

Step 5 gathers compressed bytes -- a fraction of the residual planes BASELINE.json's north star would gather
(`all_gather_planes` keeps that variant: bench.py times it as the collective of the hot path, whose output the
north star defines as the residual stream).  The backend object hides where the samples live:

    backend.scan(lo, hi)                      -> (or_word, uint64 mask words of [lo, hi); lo is a multiple of 64)
    backend.encode_range(lo, hi, file_or)     -> bytes: 43-byte header + blocks of [lo, hi)

`HipShardBackend` is the product (libsla_hip.so on this rank's GPU); the CPU tests plug the oracle in instead.
"""
import numpy as np
import torch
import torch.distributed as dist

PIECE_ALIGN = 1024          # scan pieces are cut at multiples of the search tile (and of the 64-sample mask word)


def shard_units(num_units, world, rank):
    """contiguous, balanced [lo, hi) of `num_units` for `rank` (first `num_units % world` ranks get one more)"""
    base, extra = divmod(num_units, world)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def scan_piece(num_samples, world, rank):
    """the piece of the file rank `rank` scans in step 1: [target_r, target_r+1) widened to PIECE_ALIGN on the left"""
    def cut(r):
        t = (num_samples * r + world - 1) // world
        return num_samples if r >= world else (t // PIECE_ALIGN) * PIECE_ALIGN
    return cut(rank), cut(rank + 1)


def upload_range(num_samples, world, rank, max_num_block_samples):
    """[lo, top) of the file rank `rank` needs in device memory: its scan piece plus everything its own range can reach.
    sla_hip_shard_bounds puts bounds[rank+1] at the first super-frame start at or behind the UNFLOORED target
    ceil(N*(rank+1)/world), i.e. below target + max block; the scan piece ends at that target floored to PIECE_ALIGN,
    so the range ends below piece end + PIECE_ALIGN - 1 + max block (include/sla_hip.h, step 1)."""
    lo, hi = scan_piece(num_samples, world, rank)
    return lo, min(num_samples, hi + PIECE_ALIGN - 1 + max_num_block_samples)


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


def sum_over_ranks(value, device):
    t = torch.tensor([value], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return float(t.item())


def exchange_scan(or_word, mask_piece, num_samples, device="cpu"):
    """step 2: the file's OR word and its whole mask from every rank's piece (pieces tile [0, N) in rank order)"""
    world, rank = dist.get_world_size(), dist.get_rank()
    # the OR of the ranks' words: RCCL has no bitwise reduction, so the 4 bytes are all-gathered and OR-ed here
    t = torch.tensor([or_word & 0x7FFFFFFF, or_word >> 31], dtype=torch.int32, device=device)      # (int32-safe halves)
    allt = torch.empty(2 * world, dtype=torch.int32, device=device)
    dist.all_gather_into_tensor(allt, t)
    file_or = 0
    for lo, hi in allt.cpu().numpy().reshape(world, 2):
        file_or |= int(lo) | (int(hi) << 31)
    words = [((scan_piece(num_samples, world, r)[1] - scan_piece(num_samples, world, r)[0]) + 63) // 64 for r in range(world)]
    pad = max(max(words), 1)
    mine = np.zeros(pad, np.int64)
    mine[:len(mask_piece)] = np.asarray(mask_piece, np.uint64).view(np.int64)
    out = torch.empty(world * pad, dtype=torch.int64, device=device)
    dist.all_gather_into_tensor(out, torch.from_numpy(mine).to(device))
    allw = out.cpu().numpy().reshape(world, pad)
    mask = np.concatenate([allw[r, :words[r]] for r in range(world)]).view(np.uint64)
    return file_or, mask


def exchange_counts(or_word, zero_words, device="cpu"):
    """the cheap form of step 2 (sla_hip_shard_scan_counts): 12 bytes per rank -- the file's OR word and the number of
    all-zero 64-sample mask words anywhere in it.  0 of those: no silence run can move a super-frame start, the bounds
    need no mask (sla_amd.shard_bounds(..., None, ...))."""
    world = dist.get_world_size()
    t = torch.tensor([or_word & 0x7FFFFFFF, or_word >> 31, min(int(zero_words), 0x7FFFFFFF)], dtype=torch.int32, device=device)
    allt = torch.empty(3 * world, dtype=torch.int32, device=device)
    dist.all_gather_into_tensor(allt, t)
    file_or, zeros = 0, 0
    for lo, hi, zw in allt.cpu().numpy().reshape(world, 3):
        file_or |= int(lo) | (int(hi) << 31)
        zeros += int(zw)
    return file_or, zeros


def gather_images(image, device="cpu"):
    """step 5: every rank's image on every rank (sizes, then the padded bytes; one all-gather each)"""
    world = dist.get_world_size()
    size = torch.tensor([len(image)], dtype=torch.int64, device=device)
    sizes = torch.empty(world, dtype=torch.int64, device=device)
    dist.all_gather_into_tensor(sizes, size)
    sizes = [int(v) for v in sizes.cpu()]
    pad = max(max(sizes), 1)
    mine = np.zeros(pad, np.uint8)
    mine[:len(image)] = np.frombuffer(bytes(image), np.uint8)
    out = torch.empty(world * pad, dtype=torch.uint8, device=device)
    dist.all_gather_into_tensor(out, torch.from_numpy(mine).to(device))
    allb = out.cpu().numpy().reshape(world, pad)
    return [allb[r, :sizes[r]].tobytes() for r in range(world)]


def encode_sharded(backend, num_samples, max_num_block_samples, device="cpu"):
    """steps 1-5 on this rank; returns the .sla bytes of the whole file on rank 0 (None elsewhere)"""
    import sla_amd
    world, rank = dist.get_world_size(), dist.get_rank()
    lo0, hi0 = scan_piece(num_samples, world, rank)
    mask = None
    if hasattr(backend, "scan_counts"):                       # usual case, no silence: 12 bytes per rank instead of the mask
        or_word, zero_words = backend.scan_counts(lo0, hi0)
        file_or, zeros = exchange_counts(or_word, zero_words, device)
    else:
        zeros = 1
    if zeros != 0:
        or_word, piece = backend.scan(lo0, hi0)
        file_or, mask = exchange_scan(or_word, piece, num_samples, device)
    bounds = sla_amd.shard_bounds(num_samples, max_num_block_samples, mask, world)
    image = backend.encode_range(bounds[rank], bounds[rank + 1], file_or, **({"no_silence": True} if mask is None and hasattr(backend, "scan_counts") else {}))
    images = gather_images(image, device)
    return sla_amd.shard_join(images) if rank == 0 else None


def encode_sharded_serial(backends, num_samples, max_num_block_samples):
    """the same five steps with the ranks played one after the other in ONE process (backends[r] = rank r): what the
    GPU tests run on a single-GPU box, and a way to rehearse a sharding without launching ranks"""
    import sla_amd
    world = len(backends)
    file_or, zeros, mask = 0, 1, None
    if backends and all(hasattr(b, "scan_counts") for b in backends):
        counts = [backends[r].scan_counts(*scan_piece(num_samples, world, r)) for r in range(world)]
        zeros = sum(z for _, z in counts)
        for orw, _ in counts:
            file_or |= orw
    if zeros != 0:
        scans = [backends[r].scan(*scan_piece(num_samples, world, r)) for r in range(world)]
        file_or = 0
        for orw, _ in scans:
            file_or |= orw
        mask = np.concatenate([np.asarray(m, np.uint64) for _, m in scans]) if scans else np.zeros(0, np.uint64)
    bounds = sla_amd.shard_bounds(num_samples, max_num_block_samples, mask, world)
    extra = {"no_silence": True} if (mask is None and zeros == 0) else {}
    images = [backends[r].encode_range(bounds[r], bounds[r + 1], file_or, **extra) for r in range(world)]
    return sla_amd.shard_join(images), bounds


class HipShardBackend:
    """the product's backend: this rank's GPU through libsla_hip.so.  `pcm` = the file as planar left-justified int32
    [C][N] in host memory (what SLAEncoder_EncodeWhole takes); only the rank's piece -- its scan range plus what
    `upload_range` adds behind it, which always contains the range it ends up owning -- crosses PCIe, once."""

    def __init__(self, encoder, pcm, max_num_block_samples):
        self.enc, self.pcm, self.maxb = encoder, pcm, max_num_block_samples
        self.base, self.dev = 0, None

    def _upload(self, lo, hi):
        if self.dev is not None and self.base == lo:
            return
        n = self.pcm.shape[1]
        top = min(n, hi + PIECE_ALIGN - 1 + self.maxb)      # = upload_range(): the scan piece ends at the floored target
        self.base = lo
        self.top = top
        span = max(top - lo, 1)
        self.stride = (span + 63) // 64 * 64
        self.dev = torch.zeros((self.pcm.shape[0], self.stride), dtype=torch.int32, device="cuda")
        if top > lo:
            self.dev[:, :top - lo] = torch.from_numpy(np.ascontiguousarray(self.pcm[:, lo:top])).cuda()
        torch.cuda.synchronize()

    def scan_counts(self, lo, hi):
        self._upload(lo, hi)
        return self.enc.shard_scan_counts(self.dev.data_ptr(), self.stride, hi - lo)

    def scan(self, lo, hi):
        self._upload(lo, hi)
        return self.enc.shard_scan(self.dev.data_ptr(), self.stride, hi - lo)

    def encode_range(self, lo, hi, file_or, no_silence=False):
        if hi <= lo:
            return b""                       # more ranks than super-frames: nothing to encode here
        off = lo - self.base
        if off < 0 or hi > self.top:
            raise ValueError("range [%d, %d) is not inside the uploaded piece [%d, %d)" % (lo, hi, self.base, self.top))
        self.enc.shard_analyze(self.dev.data_ptr() + 4 * off, self.stride, hi - lo, file_or, no_silence=no_silence)
        return self.enc.pack(8 * self.pcm.shape[0] * (hi - lo) + 65536, on_device=True)
