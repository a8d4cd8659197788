#!/usr/bin/env python3
"""bench.py -- encode Msamples/s of the SLA LPC+residual hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W [--config C2|C3|C4|C5] [--seconds S]

One "step" = one pass of the hot path (sla_hip_analyze_device: prepass -> partition search -> block LPC +
quantiser -> PARCOR lattice -> long-term + LMS + Rice parameter) over one batch of synthetic PCM that is already
resident in HBM when the timed region starts.

N = 1: the largest single-GPU configuration of BASELINE.json, C3 (48 kHz 24-bit stereo, 60 min, order 32) unless --config
says otherwise; the same JSON line carries `summary` ({config: [Msamples/s, ms per step, roofline.frac, 1-core CPU
Msamples/s]} for all four, among the first keys and again at the end of the line) and `other_configs` -- C2 and C5 at
their FULL length and the C4 clip batch -- each with its own roofline, verification and CPU baseline.

N > 1, one rank per GPU -- launched by torch.distributed.run, or by this script itself: `python bench.py --gpus N`
without WORLD_SIZE in the environment starts `python -m torch.distributed.run --nproc-per-node N bench.py ...` as a
child process (the parent never touches the GPU), relays rank 0's JSON line and exits with the child's code; inside the
ranks WORLD_SIZE must equal --gpus.  ONE file, its super-frames sharded over the ranks exactly as include/sla_hip.h
("one file, several GPUs") prescribes: every step scans the rank's piece, exchanges the OR word / zero-word count (and
the silence mask when there is silence) over RCCL, derives the bounds, runs the hot path on the rank's own range with
the file's OR word, and re-assembles the residual stream with one RCCL all-gather over xGMI (the north star's
collective; it travels while the next step is analysed).  --scaling weak (default): the file is N times as long as the
configuration's, per-GPU work fixed; --scaling strong: the configuration's own length (C4: --total-clips clips) split
over the N ranks (BASELINE configs 4 and 5 are fixed totals over 8 GPUs).

After the timed loop the buffers the LAST TIMED STEP left on the device are checked: block table, PARCOR bit patterns,
codes, residuals and bytes of the first super-frames against the oracle, and the packed image round-trips through the
decoder to the input PCM ("verified"); a mismatch fails the bench.

Rank 0 prints ONE JSON line (see README / DESIGN.md for the fields)."""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

# The pipeline keeps five HIP streams busy (search, block kernels, serial tail, uploads, downloads).  The ROCm runtime
# multiplexes streams onto 4 hardware queues by default, where a cross-stream event wait of one stream holds up the
# kernels of another that happens to share its queue; 8 queues give every stream its own (measured: +1..3 %).
# Must be in the environment before the HIP runtime starts.
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

import numpy as np      # noqa: E402

# name: (channels, bits, rate, seconds, order, ltm, lms, ms, window, max_block, capacity)
CONFIGS = {
    "C2": (1, 16, 48000, 600, 16, 1, 8, 0, 1, 4096, (1, 4096, 16, 1, 8)),
    "C3": (2, 24, 48000, 3600, 32, 3, 8, 1, 1, 4096, (2, 4096, 32, 3, 8)),
    "C4": (2, 16, 48000, 10, 16, 1, 8, 1, 1, 4096, (2, 4096, 16, 1, 8)),
    "C5": (8, 24, 96000, 1800, 48, 3, 8, 0, 1, 8192, (8, 8192, 48, 3, 8)),
}
WORKLOAD_NAME = {
    "C2": "synthetic 48 kHz 16-bit mono, 10 min, order-16 LPC, 4096-sample frames",
    "C3": "48 kHz 24-bit stereo, 60 min, order-32 LPC, 4096-sample frames (MS)",
    "C4": "48 kHz 16-bit stereo clip, order-16 LPC, 4096-sample frames (MS)",
    "C5": "96 kHz 24-bit 8-channel, 30 min, order-48 LPC, 8192-sample frames",
}
HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8 TB/s
FP64_VECTOR_PEAK_TFLOPS = 78.6 # MI355X FP64 vector (no dense contraction on this path: not the matrix figure); SURVEY 8(d), H8
PCIE_PEAK_GBS = 63.0           # MI355X_MICROARCH.md: host link PCIe Gen5 x16 (spec)
ALGO_BYTES_PER_SAMPLE = 8      # 4 B int32 PCM read + 4 B int32 final residual written
LCG_A, LCG_C = 1664525, 1013904223


# ---------------------------------------------------------------------------------------------- synthetic input

def lcg_jump(k):
    """(A, C) of the affine map s -> A*s + C (mod 2^32) that is k steps of BASELINE.md's LCG"""
    a, c, ra, rc = LCG_A, LCG_C, 1, 0
    while k:
        if k & 1:
            ra, rc = (a * ra) & 0xFFFFFFFF, (a * rc + c) & 0xFFFFFFFF
        a, c = (a * a) & 0xFFFFFFFF, (a * c + c) & 0xFFFFFFFF
        k >>= 1
    return ra, rc


def synth_device(torch, nch, n_total, bits, rate, lo, hi, seed=12345):
    """BASELINE.md's generator (three sines + uniform LCG noise, rounded to `bits`, left-justified) for samples [lo, hi)
    of an n_total-sample file, made ON the device (an hour of 24-bit stereo is minutes of numpy): int32 [nch][hi - lo]"""
    n = hi - lo
    out = torch.empty((nch, max(n, 1)), dtype=torch.int32, device="cuda")
    if n <= 0:
        return out[:, :0]
    full = float(1 << (bits - 1))
    mask = (1 << 32) - 1
    for ch in range(nch):
        chunk = 1 << 24
        for c0 in range(0, n, chunk):
            m = min(chunk, n - c0)
            # LCG state of sample index (lo + c0 + j), j = 0..m-1, on channel ch: seed advanced by ch*n_total + index + 1 steps
            ja, jc = lcg_jump(ch * n_total + lo + c0)
            s0 = (ja * seed + jc) & mask
            a = torch.full((m,), LCG_A, dtype=torch.int64, device="cuda")
            pw = torch.cumprod(a, 0) & mask                                     # A^(j+1) mod 2^32 (wraps mod 2^64 first: consistent)
            geo = (torch.cumsum(torch.cat([torch.ones(1, dtype=torch.int64, device="cuda"), pw[:-1]]), 0)) & mask   # 1 + A + .. + A^j
            s = (pw * s0 + geo * LCG_C) & mask
            noise = ((s >> 8).to(torch.float64) / float(1 << 24) - 0.5) * (2.0 * 0.02)
            t = (torch.arange(lo + c0, lo + c0 + m, dtype=torch.float64, device="cuda")) / float(rate)
            x = (0.35 * torch.sin(2 * np.pi * 220.0 * (ch + 1) * t) + 0.2 * torch.sin(2 * np.pi * 1333.7 * t + ch)
                 + 0.1 * torch.sin(2 * np.pi * 5011.3 * t) + noise)
            q = torch.clamp(torch.round(x * full), -full, full - 1).to(torch.int64)
            out[ch, c0:c0 + m] = (q << (32 - bits)).to(torch.int32)
            del a, pw, geo, s, noise, t, x, q
    return out


def upload_bytes_per_sample(bits):
    """bytes per sample that pageable host PCM crosses the bus with: staged as int16 up to 16 significant bits, as three
    bytes up to 24 (option upload24, on by default), as the int32 it is above that"""
    return 2 if bits <= 16 else (3 if bits <= 24 else 4)


# ---------------------------------------------------------------------------------------------- CPU baseline

def cpu_info():
    model = ""
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                model = line.split(":", 1)[1].strip()
                break
    except OSError:
        pass
    try:
        usable = len(os.sched_getaffinity(0))
    except (AttributeError, OSError):
        usable = os.cpu_count() or 1
    return model, os.cpu_count() or 1, usable


MAX_CPU_WORKERS = 16                  # host share of one GPU on the pool's boxes (they expose all 256 hardware threads)
CPU_SAMPLE_PER_CHANNEL = 1 << 21      # samples per channel of the CPU legs' input: a few hundred MB of numpy at most, per process


def _cpu_sample(S, cfg, seed=12345):
    nch, bits, rate = CONFIGS[cfg][0], CONFIGS[cfg][1], CONFIGS[cfg][2]
    if cfg == "C4":
        return [S.synth_pcm(nch, int(rate * CONFIGS[cfg][3]), bits, rate, seed=4000 + k) for k in range(4)]     # four 10-second clips
    return [S.synth_pcm(nch, CPU_SAMPLE_PER_CHANNEL // max(nch // 2, 1), bits, rate, seed=seed)]


def _cpu_loop(checker, p, subs, budget_s):
    """encode the sample over and over until `budget_s` of wall time is used (at least once): (samples x channels, seconds)"""
    samples, t0 = 0, time.perf_counter()
    while True:
        for sub in subs:
            ret, _ = checker.encode_whole(p, sub)
            if ret != 0:
                return 0, 0.0
            samples += sub.shape[0] * sub.shape[1]
        if time.perf_counter() - t0 >= budget_s:
            return samples, time.perf_counter() - t0


def _cpu_worker(args):
    """one process of the all-cores figure: the checker's EncodeWhole, pinned to its own core, on its own copy of the sample"""
    (cfg, seed, cpu, budget_s) = args
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import slalibs as S
    nch, bits, rate, _, order, ltm, lms, ms, win, maxb, cap = CONFIGS[cfg]
    try:
        os.sched_setaffinity(0, {cpu})
    except (AttributeError, OSError):
        pass
    p = S.make_params(nch, bits, rate, order, ltm, lms, ms, win, maxb, cap=cap)
    ref = S.ref()
    checker = ref if ref is not None else S.oracle()
    subs = _cpu_sample(S, cfg, seed)
    return _cpu_loop(checker, p, subs, budget_s)


def cpu_baseline(S, cfg, budget_s, all_cores_seconds):
    """The unmodified reference (oracle/_ref, kind "reference") or the oracle restatement ("port") on ONE core it is pinned
    to (= taskset -c), encoding a bounded sample of the configuration's workload again and again for `budget_s` seconds; with
    all_cores_seconds > 0 also every usable core at once (one process + one encoder per core).
    Runs BEFORE this process touches the GPU: no process is ever started from a GPU-initialised one, and the sample is small
    (the first version handed every worker minutes of audio: 16 workers x 6 GB of numpy temporaries took the box down)."""
    nch, bits, rate, _, order, ltm, lms, ms, win, maxb, cap = CONFIGS[cfg]
    p = S.make_params(nch, bits, rate, order, ltm, lms, ms, win, maxb, cap=cap)
    ref = S.ref()
    checker, kind = (ref, "reference") if ref is not None else (S.oracle(), "port")
    model, nproc, usable = cpu_info()
    subs = _cpu_sample(S, cfg)
    pinned, before = None, None
    try:
        before = os.sched_getaffinity(0)
        pinned = min(before)
        os.sched_setaffinity(0, {pinned})
    except (AttributeError, OSError):
        before = None
    try:
        samples, cpu_s = _cpu_loop(checker, p, subs, budget_s)
    finally:
        if before is not None:
            os.sched_setaffinity(0, before)
    assert samples > 0
    per_pass = sum(x.shape[0] * x.shape[1] for x in subs)
    out = {"value": round(samples / cpu_s / 1e6, 3), "unit": "Msamples/s", "cores": 1, "kind": kind, "seconds": round(cpu_s, 2),
           "pinned_to_cpu": pinned, "cpu_model": model, "nproc": nproc, "usable_cpus": usable,
           "sample": "%s of the same workload (%d samples x channels), encoded %d times: full single-thread EncodeWhole incl. the bit-pack (%s)"
                     % ("four 10 s clips, one call each" if cfg == "C4" else "the first %d samples per channel" % subs[0].shape[1],
                        per_pass, samples // per_pass, "unmodified reference, oracle/_ref" if kind == "reference" else "oracle restatement")}
    if all_cores_seconds > 0 and usable > 1:
        # plain child processes of this (GPU-free) interpreter: `bench.py --cpu-worker ...`, one per usable core
        import subprocess
        cpus = sorted(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else list(range(usable))
        cpus = cpus[:MAX_CPU_WORKERS]                 # a one-GPU box's share of the host, however many CPUs it shows
        t1 = time.perf_counter()
        procs = [subprocess.Popen([sys.executable, os.path.abspath(__file__), "--cpu-worker", cfg, str(777 + k), str(cpus[k]), str(all_cores_seconds)],
                                  stdout=subprocess.PIPE, stderr=subprocess.DEVNULL) for k in range(len(cpus))]
        res = []
        for pr in procs:
            try:
                so, _ = pr.communicate(timeout=all_cores_seconds * 6 + 120)
                res.append(tuple(json.loads(so.decode().strip().splitlines()[-1])))
            except Exception:
                pr.kill()
                res.append((0, 0.0))
        wall = time.perf_counter() - t1
        if all(r[0] > 0 for r in res):
            out["all_cores"] = {"value": round(sum(r[0] / r[1] for r in res) / 1e6, 3), "unit": "Msamples/s",
                                "cores": len(cpus), "seconds": round(max(r[1] for r in res), 2), "wall_incl_startup": round(wall, 2),
                                "sample": "one process + one encoder per core of this GPU's host share (%d of %d visible CPUs), each on its own copy of the single-core sample, same time window" % (len(cpus), usable)}
    return out


# ---------------------------------------------------------------------------------------------- one configuration

def run_config(torch, sla_amd, S, cfg, args, rank, world, primary, cpu_results=None):
    from sla_amd import dist as sdist
    dist = None
    # the collectives of the N > 1 path run at N = 1 too under --force-collectives: the same calls on a one-rank communicator
    dist_on = (world > 1) or (args.force_collectives and primary)
    if dist_on:
        import torch.distributed as dist
    nch, bits, rate, seconds, order, ltm, lms, ms, win, maxb, cap = CONFIGS[cfg]
    if primary and args.seconds is not None:
        seconds = args.seconds
    n_cfg = int(rate * seconds)
    dev_comm = "cuda" if args.backend == "nccl" else "cpu"
    batch = None
    host_pcm = None
    if cfg == "C4":
        # a batch of clips per step (BASELINE config 3: 1000 clips over 8 GPUs), laid out back to back on
        # 1024-sample boundaries and analysed in ONE pipeline pass (sla_hip_analyze_batch_device)
        clip_n, tile = n_cfg, 1024
        pitch = (clip_n + tile - 1) // tile * tile
        # weak: --clips per GPU and step; strong: --total-clips round-robin over the ranks (C4 = 1000 clips over 8 GPUs)
        nclips = args.clips if args.scaling == "weak" else len(range(rank, args.total_clips, world))
        distinct = [S.synth_pcm(nch, clip_n, bits, rate, seed=4000 + 16 * rank + k) for k in range(min(16, max(nclips, 1)))]
        clips = [distinct[k % len(distinct)] for k in range(nclips)]
        batch = {"starts": np.arange(nclips, dtype=np.uint32) * pitch, "lens": np.full(nclips, clip_n, np.uint32),
                 "clips": clips, "span": max(nclips, 1) * pitch, "count": nclips}
        n_own = clip_n * nclips                          # samples per channel that are audio
        stride = batch["span"]
        d_pcm = torch.zeros((nch, stride), dtype=torch.int32, device="cuda")
        for k, c in enumerate(clips):
            d_pcm[:, k * pitch:k * pitch + clip_n] = torch.from_numpy(c).cuda()
        n_file, lo0, hi0, base = n_own, 0, n_own, 0
    else:
        # one file of world x the configuration's length (weak) or of the configuration's own length (strong); this
        # rank holds its scan piece plus one maximum block
        n_file = n_cfg * world if args.scaling == "weak" else n_cfg
        lo0, hi0 = sdist.scan_piece(n_file, world, rank)
        top = sdist.upload_range(n_file, world, rank, maxb)[1] if world > 1 else n_file
        base = lo0
        span = top - lo0
        # the same plane stride on every rank (the residual planes are all-gathered as they are)
        widest = max(((sdist.upload_range(n_file, world, r, maxb)[1] if world > 1 else n_file) - sdist.scan_piece(n_file, world, r)[0])
                     for r in range(world))
        stride = (widest + 63) // 64 * 64
        d_pcm = torch.zeros((nch, stride), dtype=torch.int32, device="cuda")
        if cfg == "C2" and world == 1:
            host_pcm = S.synth_pcm(nch, n_file, bits, rate, seed=12345)           # the numpy generator of BASELINE.md, as in round 1
            d_pcm[:, :span] = torch.from_numpy(host_pcm).cuda()
        else:
            d_pcm[:, :span] = synth_device(torch, nch, n_file, bits, rate, lo0, top)
        n_own = hi0 - lo0                                # refined per step by the bounds (differs by < one block)
    torch.cuda.synchronize()

    # N > 1 over RCCL: the all-gather of one step's residual planes travels while the next step is analysed into
    # a second set of planes (two sets take turns), so the collective over xGMI and the kernels overlap
    overlap = (dist_on and args.backend == "nccl" and not args.sync_gather)
    nbuf = 2 if overlap else 1
    d_lat = [torch.zeros((nch, stride), dtype=torch.int32, device="cuda") for _ in range(nbuf)]
    d_fin = [torch.zeros((nch, stride), dtype=torch.int32, device="cuda") for _ in range(nbuf)]
    gathered = [torch.empty((world, nch, stride), dtype=torch.int32, device="cuda") for _ in range(nbuf)] if dist_on else None
    works = [None] * nbuf

    enc = sla_amd.Encoder(*cap)
    enc.set_wave_format(nch, bits, rate)
    enc.set_encode_parameter(order, ltm, lms, ms, win, maxb)
    enc.bind_residual_planes(d_lat[0].data_ptr(), d_fin[0].data_ptr(), stride)
    torch.cuda.synchronize()

    state = {"step": 0, "own": (lo0, hi0), "file_or": 0}
    span_ms = np.zeros(4)       # on-device execution time of k_lpc_blocks, k_lattice, k_ltm_acf, k_tail (summed over steps)

    def settle(b):
        if works[b] is not None:
            works[b].wait()
            torch.cuda.current_stream().synchronize()
            works[b] = None

    def step():
        b = state["step"] % nbuf
        state["step"] += 1
        settle(b)                                             # the planes of two steps ago have been gathered
        if nbuf > 1:
            enc.bind_residual_planes(d_lat[b].data_ptr(), d_fin[b].data_ptr(), stride)
        if batch is not None:
            t, _ = enc.analyze_batch_device(d_pcm.data_ptr(), stride, batch["span"], batch["starts"], batch["lens"])
        elif not dist_on:
            t = enc.analyze_device(d_pcm.data_ptr(), stride, n_file)
        else:
            # one file over the ranks (include/sla_hip.h): scan, exchange, bounds, the hot path on the own range
            # (12 bytes per rank when no piece has an all-zero mask word -- no silence run can move a super-frame start then --
            # and the 1-bit mask of the whole file only otherwise)
            orw, zero_words = enc.shard_scan_counts(d_pcm.data_ptr(), stride, hi0 - lo0)
            file_or, zeros = sdist.exchange_counts(orw, zero_words, dev_comm)
            mask = None
            if zeros != 0:
                orw, piece = enc.shard_scan(d_pcm.data_ptr(), stride, hi0 - lo0)
                file_or, mask = sdist.exchange_scan(orw, piece, n_file, dev_comm)
            bounds = sla_amd.shard_bounds(n_file, maxb, mask, world)
            lo, hi = bounds[rank], bounds[rank + 1]
            state["own"], state["file_or"] = (lo, hi), file_or
            t = enc.shard_analyze(d_pcm.data_ptr() + 4 * (lo - base), stride, hi - lo, file_or, no_silence=(mask is None))
        span_ms[:] += np.array(enc.last_kernel_ms())
        if dist_on:
            if overlap:
                works[b] = sdist.all_gather_planes(d_fin[b], gathered[b], async_op=True)[1]
            elif args.backend == "nccl":
                sdist.all_gather_planes(d_fin[b], gathered[b])   # RCCL over xGMI: re-assemble the residual stream
            else:                                             # rehearsal backend: stage through the host
                gathered[b].copy_(sdist.all_gather_planes(d_fin[b].cpu()))
        return t

    # the collector is switched off for the timed loop (as timeit does): a generation-2 pass over this process's torch and
    # numpy objects is a pause of several milliseconds in a loop of 1.2 ms steps
    import gc
    gc.collect()
    gc.disable()
    # the FIRST analysis on this fresh handle (buffers allocated, search tables built and uploaded, window tables made,
    # nothing kept from an earlier file): one of the warm-up steps when there is one, an extra untimed step otherwise
    torch.cuda.synchronize()
    tc = time.perf_counter()
    step()
    torch.cuda.synchronize()
    cold_call_ms = (time.perf_counter() - tc) * 1e3
    for _ in range(max(args.warmup - 1, 0)):
        step()
    for b in range(nbuf):
        settle(b)
    torch.cuda.synchronize()
    if dist_on:
        dist.barrier()
    kernel_ms = np.zeros(12)
    span_ms[:] = 0.0
    step_wall = []
    t0 = time.perf_counter()
    for _ in range(args.steps):
        ts = time.perf_counter()
        kernel_ms += np.array(step())
        step_wall.append(time.perf_counter() - ts)
    for b in range(nbuf):
        settle(b)                                             # every collective of the timed steps has landed
    torch.cuda.synchronize()
    if dist_on:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    gc.enable()
    if dist_on:
        elapsed = sdist.max_over_ranks(elapsed, dev_comm)
    kernel_ms /= max(args.steps, 1)
    span_ms /= max(args.steps, 1)

    if batch is not None and dist_on:
        n_all = int(sdist.sum_over_ranks(float(n_own), dev_comm))      # samples per channel all ranks analysed per step
    else:
        n_all = n_file if batch is None else n_own
    total_samples = float(n_all) * nch * args.steps
    value = total_samples / elapsed / 1e6
    own_lo, own_hi = state["own"]
    n_last = (own_hi - own_lo) if batch is None else n_own

    out = {
        "value": round(value, 3), "unit": "Msamples/s", "ms_per_step": round(elapsed / max(args.steps, 1) * 1e3, 3),
        # value and ms_per_step are the contract's mean over exactly K steps; the per-step spread beside them (one call
        # per step, synchronous at N = 1): a host stall -- the boxes' CPU quota, see DESIGN section 7 -- shows as max >> median
        "ms_per_step_median": round(float(np.median(step_wall)) * 1e3, 3) if step_wall else None,
        "ms_per_step_max": round(float(np.max(step_wall)) * 1e3, 3) if step_wall else None,
        "cold_call_ms": round(cold_call_ms, 3),
        "config": {"workload": WORKLOAD_NAME[cfg] + (", batch of %d clips on this GPU in one pass" % batch["count"] if batch else "")
                               + (", ONE file of %d x that length sharded over %d ranks" % (world, world) if world > 1 and batch is None and args.scaling == "weak" else "")
                               + (", that ONE file sharded over %d ranks" % world if world > 1 and batch is None and args.scaling == "strong" else ""),
                   "name": cfg, "channels": nch, "bits": bits, "rate": rate,
                   "seconds": seconds * (world if batch is None and args.scaling == "weak" else 1),
                   "parcor_order": order, "longterm_order": ltm, "lms_order": lms,
                   "max_block_samples": maxb, "samples_per_step_per_gpu": int((n_file // world if batch is None else n_own) * nch),
                   "samples_per_step_all_gpus": int(n_all) * nch,
                   "parallelism": ("super-frames of one file sharded over %d GPUs: OR all-reduce + mask all-gather, then one RCCL all-gather "
                                   "of the residual planes%s" % (world, " overlapped with the next step" if overlap else ""))
                                  if world > 1 else ("1 GPU, the collectives of the N > 1 path on a one-rank communicator: count exchange + all-gather of the residual planes%s"
                                                     % (" overlapped with the next step" if overlap else "") if dist_on else "1 GPU")},
    }
    if rank != 0:
        enc.close()
        return out

    # ---- roofline of the dominant kernel (HIP-event durations measured inside the library, on the stream the
    #      kernels run on) ----------------------------------------------------------------------------------
    nchunks = max(int(round(kernel_ms[9])), 1)
    exact_search = kernel_ms[11] > 0.5      # partition search ran on tile sums
    cert_on, blocks_exact = enc.last_block_cert()
    # the block stage: k_acf_blocks + k_blocks_finish (+ the exact chain kernels for what did not certify), or
    # k_lpc_blocks + k_blocks_finish with option block_cert = 0
    blk = "k_acf_blocks" if cert_on else "k_lpc_blocks"
    ev = {blk: kernel_ms[2], "k_lattice": kernel_ms[3], "k_ltm_acf": kernel_ms[8], "k_tail": kernel_ms[4]}
    dev = dict(zip((blk, "k_lattice", "k_ltm_acf", "k_tail"), span_ms))
    kernels = {"k_prepass": (kernel_ms[0], 1)}
    tail_launches = max(int(enc.last_counters()[4]), 1)         # one k_tail for the whole file unless the handle was told otherwise
    for name in ev:
        kernels[name] = (ev[name], tail_launches if name == "k_tail" else nchunks)      # HIP events on the kernel's own stream, as the contract asks
    search_name = ("k_acf_tiles+k_search_finish" + ("+k_lpc" if kernel_ms[10] > 0 else "")) if exact_search else "k_lpc"
    kernels[search_name] = (kernel_ms[1], nchunks)               # one event pair spans the kernels of the search stage
    dom = max(kernels, key=lambda k: kernels[k][0])
    t_step, launches = kernels[dom]
    launch_ms = t_step / launches
    algo_bytes = float(n_last) * nch * ALGO_BYTES_PER_SAMPLE / launches      # per launch: 8 B x the samples x channels it covers
    achieved = algo_bytes / (launch_ms * 1e-3) / 1e9 if launch_ms > 0 else 0.0
    # ---- secondary bounds: FP64 vector rate of what the kernels EXECUTE ----------------
    # per sample x channel: tile-sum search 2 x lags flop (one FMA per lag; lags = order + 1 rounded up to a multiple of 4:
    # 12/20/36/52); block stage the same on the windowed samples + 8 flop of staging (exact chain kernel: 1.5 x order + 2);
    # long-term autocorrelation: two radix-2 transforms of npts = capacity complex points, 10 flop per butterfly,
    # + 2 x 14 flop per point of recombination and power spectrum, per block of max_block samples; lattice, LMS tail,
    # prepass and the Rice stages are integer kernels (no FP64)
    lags = float(sla_amd.lib().sla_hip_search_exact_lags(order)) or float(order + 1)
    npts_fft = 1
    while npts_fft < cap[1]:
        npts_fft <<= 1
    lg = npts_fft.bit_length() - 1
    flop_per_sc = {blk: (2.0 * lags + 8.0) if cert_on else (1.5 * order + 2.0),
                   search_name: (2.0 * lags if exact_search else 1.5 * order * (4 if maxb <= 4096 else 14)),
                   "k_ltm_acf": (2.0 * 10.0 * (npts_fft / 2) * lg + 28.0 * npts_fft) / float(maxb)}
    sc_step = float(n_last) * nch
    path_flop = sum(flop_per_sc.values()) * sc_step
    # ---- integer-VALU bound of the integer kernels ----------------------------------------------
    # k_lattice_groups: 2 terms R(k v) per sample and stage; a stage runs in the form its operand bound allows (sla_kernels.hip):
    # 24-bit material with its 8-bit coefficients certifies the two-instruction form H (high dword of v_mad_i64_i32 + v_sub_u32)
    # in every stage, 16-bit material H or S (v_mad_i32_i24 + SDWA subtract).  tests/tools/ubench_lattice.hip measured, at eight
    # waves per SIMD on all 256 CUs, the sustained rate of each form on registers alone (profiles/r4_ubench_lattice_8w.txt):
    # H 18.48, S 17.90, round 3's four-instruction form 9.61 T terms/s; plain v_add_u32 61.3 T lane-ops/s = 0.78 of the
    # nominal 78.6 T (256 CU x 4 SIMD x 32 lanes x 2.4 GHz: the clock under load is ~1.9 GHz), so a term of form H costs 3.3 add slots.
    # int_valu_frac = terms x add-slots per term / launch time / 78.6 T; int_valu_sustained_frac = terms/s / the ubench's rate.
    # k_tailk: ~33 wave instructions per sample for 16 jobs (K = 2; many jobs) or ~27 for 8 (K = 1): lane-ops = that x 64 lanes.
    VALU_PEAK = 78.6e12
    int_model = {}
    terms = sc_step * order * 2.0 - sc_step                       # the last stage has one term per sample
    slots_per_term, sustained = (3.32, 18.48e12) if bits > 16 else (3.4, 17.9e12)
    int_model["k_lattice"] = {"lane_ops": terms * slots_per_term, "sustained_units": terms, "sustained_rate": sustained,
                              "what": "2 terms (k v + 2^14) >> 15 per sample and stage, %.2f add-slots per term (form %s)" % (slots_per_term, "H" if bits > 16 else "H / S")}
    many = (n_last * nch // maxb) * lms // 64 > 2048              # (the launcher's switch: two taps per lane beyond 2048 one-tap waves)
    per_sample_job = (33.0 / 16.0) if many else (27.0 / 8.0)
    int_model["k_tail"] = {"lane_ops": sc_step * per_sample_job * 64.0, "sustained_units": None, "sustained_rate": None,
                           "what": "%.2f wave instructions per sample and job (K = %d taps per lane) x 64 lanes" % (per_sample_job, 2 if many else 1)}
    out["roofline"] = {"bound": "hbm", "kernel": {"k_tail": "k_tailk", "k_ltm_acf": "k_ltm_acf2", "k_lattice": "k_lattice_groups"}.get(dom, dom), "achieved": round(achieved, 2),
                       "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 5),
                       "path_frac": round(sc_step * ALGO_BYTES_PER_SAMPLE / (elapsed / max(args.steps, 1)) / 1e9 / HBM_PEAK_GBS, 5),
                       "fp64_vector_frac": (round(flop_per_sc[dom] * sc_step / launches / (launch_ms * 1e-3) / 1e12 / FP64_VECTOR_PEAK_TFLOPS, 5)
                                            if dom in flop_per_sc and launch_ms > 0 else None),
                       "int_valu_frac": (round(int_model[dom]["lane_ops"] / launches / (launch_ms * 1e-3) / VALU_PEAK, 5)
                                         if dom in int_model and launch_ms > 0 else None),
                       "int_valu_sustained_frac": (round(int_model[dom]["sustained_units"] / launches / (launch_ms * 1e-3) / int_model[dom]["sustained_rate"], 5)
                                                   if dom in int_model and int_model[dom]["sustained_rate"] and launch_ms > 0 else None),
                       "int_valu_model": int_model[dom]["what"] if dom in int_model else None,
                       "int_valu_frac_by_kernel": {{"k_tail": "k_tailk", "k_lattice": "k_lattice_groups"}[k]:
                                                   round(v["lane_ops"] / max(kernels[k][1], 1) / (kernels[k][0] / max(kernels[k][1], 1) * 1e-3) / VALU_PEAK, 5)
                                                   for k, v in int_model.items() if kernels[k][0] > 0},
                       "path_fp64_vector_frac": round(path_flop / (elapsed / max(args.steps, 1)) / 1e12 / FP64_VECTOR_PEAK_TFLOPS, 5),
                       "fp64_flop_per_sample_channel": {k: round(v, 1) for k, v in flop_per_sc.items()},
                       "fp64_peak_tflops": FP64_VECTOR_PEAK_TFLOPS,
                       "secondary_note": "int_valu_frac = integer lane-ops of the dominant kernel, each at its measured issue cost in v_add_u32 slots "
                                         "(tests/tools/ubench_lattice.hip, profiles/r4_ubench_lattice_8w.txt) / launch time / 78.6 T lane-ops/s; int_valu_sustained_frac = "
                                         "against what the same instruction mix sustains on registers alone; None for an FP64 kernel; "
                                         "path_frac = 8 B x samples x channels of one step / ms_per_step / 8 TB/s; fp64_vector_frac = executed FP64 "
                                         "flop of the dominant kernel (model in fp64_flop_per_sample_channel; None for an integer kernel) / its "
                                         "launch time / 78.6 TFLOP/s; path_fp64_vector_frac = the FP64 flop of all stages / ms_per_step / 78.6 TFLOP/s",
                       "traffic": None, "kernel_ms": round(float(launch_ms), 4), "launches_per_step": launches,
                       "algorithmic_bytes_per_launch": algo_bytes,
                       "kernel_ms_running": round(float(dev[dom] / launches), 4) if dom in dev and dev[dom] > 0 else None,
                       "note": "kernel_ms = average duration of one launch of this kernel inside the timed region, between HIP events "
                               "on the stream it is launched on (what rocprofv3 --kernel-trace --stats of the same command shows, "
                               "profiles/r3_kernel_stats_*.csv; both include the time a launch shares the SIMDs with, or waits "
                               "behind, the next chunk's kernels on the other streams); kernel_ms_running = first workgroup in to "
                               "last workgroup out on the device's 100 MHz clock, i.e. without that wait"}
    try:       # HBM traffic per launch of that kernel from the committed rocprofv3 --pmc passes of this configuration
        pmc = json.load(open(os.path.join(ROOT, "profiles", "pmc_traffic_%s.json" % cfg.lower())))
        pk = dom.split("+")[0]
        for alias in (pk + "2", pk + "k", pk + "_groups"):    # the stage "k_ltm_acf" runs k_ltm_acf2, "k_tail" k_tailk, "k_lattice" k_lattice_groups
            if pk not in pmc["bytes_per_launch"] and alias in pmc["bytes_per_launch"]:
                pk = alias
        if world == 1 and pk in pmc["bytes_per_launch"]:
            rec = pmc["bytes_per_launch"][pk]
            if "bytes_per_sample_channel" in rec:
                # the counters were collected on a shorter cut of the same workload: every kernel of the path moves a fixed
                # number of bytes per sample and channel (blocks are independent), so the per-launch figure scales with the
                # samples x channels a launch covers
                out["roofline"]["traffic"] = int(round(rec["bytes_per_sample_channel"] * sc_step / launches))
                out["roofline"]["traffic_source"] = pmc["source"] + "; scaled by samples x channels per launch"
            else:
                same_size = (str(pmc.get("seconds", "full")) == "full" and seconds == CONFIGS[cfg][3]) or str(pmc.get("seconds")) == str(int(seconds))
                if same_size:
                    out["roofline"]["traffic"] = rec["total"]
                    out["roofline"]["traffic_source"] = pmc["source"]
    except (OSError, KeyError, ValueError):
        pass
    out["block_certificate"] = {"on": bool(cert_on), "pairs_redone_by_exact_kernels": int(blocks_exact)}
    out["kernel_ms_on_device"] = {k: round(float(v), 4) for k, v in dev.items()}
    out["stage_ms"] = {"k_prepass": round(float(kernel_ms[0]), 4),
                       ("search_tile_sums" if exact_search else "k_lpc_search"): round(float(kernel_ms[1]), 4),
                       "search_groups_rerun_as_chains": round(float(kernel_ms[10]), 2),
                       "block_stage(" + blk + "+k_blocks_finish)": round(float(kernel_ms[2]), 4), "k_lattice": round(float(kernel_ms[3]), 4),
                       "k_tail": round(float(kernel_ms[4]), 4), "k_ltm_acf": round(float(kernel_ms[8]), 4), "host_plan": round(float(kernel_ms[5]), 4),
                       "host_longterm": round(float(kernel_ms[6]), 4), "analyze_total": round(float(kernel_ms[7]), 4)}

    # ---- verification of what the last TIMED step left in the encoder and on the device -----------------------
    o = S.oracle()
    p = S.make_params(nch, bits, rate, order, ltm, lms, ms, win, maxb, cap=cap)
    ver = {"checked_against": "oracle (CPU restatement pinned to the unmodified reference, tests/test_oracle_vs_ref.py)"}
    if batch is not None:
        tr = enc.trace()
        nb = tr.num_blocks
        ok, checked = True, 0
        for k in range(min(2, batch["count"])):
            ret, want, to = o.encode_trace(p, batch["clips"][k])
            start = int(batch["starts"][k])
            sel = [b for b in range(nb) if start <= tr.blk_start[b] < start + int(batch["lens"][k])]
            ok &= (ret == 0 and len(sel) == to.num_blocks)
            for j, b in enumerate(sel[:to.num_blocks]):
                ok &= (int(tr.blk_nsmpl[b]) == int(to.blk_nsmpl[j]) and int(tr.blk_type[b]) == int(to.blk_type[j]))
                if to.blk_type[j] == 0:
                    s0, s1, ln = int(tr.blk_start[b]), int(to.blk_start[j]), int(to.blk_nsmpl[j])
                    ex = tr.parcor_exact[b].astype(bool)      # exact chain kernel: bit patterns; certified route: the codes decide
                    ok &= bool(np.array_equal(tr.parcor[b].view(np.uint64)[ex], to.parcor[j].view(np.uint64)[ex])
                               and np.all(np.abs(tr.parcor[b][~ex] - to.parcor[j][~ex]) <= 2.0 ** -9)
                               and np.array_equal(tr.code[b], to.code[j]) and np.array_equal(tr.kint[b], to.kint[j]) and np.array_equal(tr.rice_init[b], to.rice_init[j])
                               and np.array_equal(tr.res_final[:, s0:s0 + ln], to.res_final[:, s1:s1 + ln]))
                checked += 1
        ver.update({"blocks_compared": checked, "fields": "block table, PARCOR (bit patterns where the exact kernel ran, a quarter quantisation step where certified), codes, Rice parameters, final residual of the first 2 clips"})
    else:
        own_n = own_hi - own_lo
        frames = min(20 if maxb <= 4096 else 8, max(own_n // maxb - 1, 1))
        m = min(frames * maxb, own_n)
        sub = np.ascontiguousarray(d_pcm[:, own_lo - base:own_lo - base + m].cpu().numpy())
        file_or = state["file_or"]
        if dist_on:
            ntz = (file_or & -file_or).bit_length() - 1 if file_or else 32
            lshift = bits - (32 - ntz) if file_or else 0
            ret, want = o.encode_range(p, sub, lshift)
            to = None
        else:
            ret, want, to = o.encode_trace(p, sub)
        image = enc.pack(min(4 * nch * own_n + 65536, 0xFFFFFFF0), on_device=True)     # the bit-pack of the TIMED analysis
        tr = enc.trace(want_residuals=False, max_blocks=own_n // 2048 + 8)
        ok = (ret == 0)
        # blocks are independent: the oracle's encode of the first frames must reproduce the first bytes of the image
        # (the prefix's last super-frame can end the "file" differently, so compare up to the block before it)
        if m == own_n:
            ok &= (image[43:] == want[43:])
            nblk = tr.num_blocks
        else:
            nblk, pos, off = 0, 0, 43
            while nblk < tr.num_blocks and pos + int(tr.blk_nsmpl[nblk]) <= m - maxb:
                pos += int(tr.blk_nsmpl[nblk]); off += int(tr.blk_bytes[nblk]); nblk += 1
            ok &= (nblk > 0 and image[43:off] == want[43:off])
        if to is not None:
            ok &= bool(S.parcor_same(tr, to, nblk)
                       and np.array_equal(tr.code[:nblk], to.code[:nblk]) and np.array_equal(tr.kint[:nblk], to.kint[:nblk]) and np.array_equal(tr.rice_init[:nblk], to.rice_init[:nblk])
                       and np.array_equal(tr.blk_nsmpl[:nblk], to.blk_nsmpl[:nblk]))
        ver.update({"blocks_compared": int(nblk), "fields": "bytes of the first blocks (headers, Rice bodies, CRC16)"
                    + (", PARCOR (bit patterns where the exact kernel ran, a quarter quantisation step where certified), codes, Rice parameters" if to is not None else "")})
        # two more windows, in the middle and at the end of the rank's range, against the oracle's encode of the same samples
        # (a range that starts on a super-frame start reproduces the file's own blocks: include/sla_hip.h, one file, several GPUs)
        if m < own_n and tr.num_blocks > 3:
            lsh = tr.offset_lshift
            starts = tr.blk_start[:tr.num_blocks].astype(np.int64)
            byte_off = 43 + np.concatenate([[0], np.cumsum(tr.blk_bytes[:tr.num_blocks].astype(np.int64))])
            sf = np.nonzero((starts - starts[0]) % maxb == 0)[0]                  # blocks that open a super-frame (no silence in this signal)
            extra = 0
            for b0 in (int(sf[len(sf) // 2]), int(sf[max(len(sf) - frames, 0)])):
                s0 = int(starts[b0]) - int(starts[0])
                mm = min(frames * maxb, own_n - s0)
                subw = np.ascontiguousarray(d_pcm[:, own_lo - base + s0:own_lo - base + s0 + mm].cpu().numpy())
                retw, wantw = o.encode_range(p, subw, lsh)
                b1, pos = b0, 0
                to_end = (s0 + mm == own_n)
                while b1 < tr.num_blocks and (to_end or pos + int(tr.blk_nsmpl[b1]) <= mm - maxb):
                    pos += int(tr.blk_nsmpl[b1]); b1 += 1
                seg = image[int(byte_off[b0]):int(byte_off[b1])]
                ok &= bool(retw == 0 and b1 > b0 and seg == wantw[43:43 + len(seg)] and (not to_end or len(wantw) == 43 + len(seg)))
                extra += b1 - b0
            ver["blocks_compared_mid_and_end"] = int(extra)
        # ... and the whole image decodes back to the PCM this rank analysed (size-independent property at full size)
        if lms in (4, 8, 16, 32) and maxb <= 16384:
            dec = sla_amd.Decoder(nch, maxb, order, ltm, lms)
            d_img = torch.frombuffer(bytearray(image), dtype=torch.uint8).cuda()
            d_out = torch.zeros((nch, stride), dtype=torch.int32, device="cuda")
            rc, nsm = dec.decode_device(np.frombuffer(image, np.uint8), d_img.data_ptr(), d_out.data_ptr(), stride)
            same = bool(rc == 0 and nsm == own_n and torch.equal(d_out[:, :own_n], d_pcm[:, own_lo - base:own_lo - base + own_n]))
            ok &= same
            ver["round_trip_identical"] = same
            ver["round_trip_samples"] = int(own_n) * nch
            dec.close()
            del d_img, d_out
    ver["ok"] = bool(ok)
    out["verified"] = bool(ok)
    out["verification"] = ver
    if not ok:
        print(json.dumps({"error": "verification failed", "config": cfg, "verification": ver}), file=sys.stderr, flush=True)
        raise SystemExit(3)

    # ---- what repeating ONE file on ONE handle spares a step: the search tables of a file without silence
    #      are kept for the next file of the same shape, and for short files the searches are launched behind the prepass
    #      on the guess that the file looks like the last one.  `value` is that warm, same-shape steady state (the contract's
    #      K steps after W warm-ups); beside it: the same steps with nothing kept (every step builds and uploads its tables and
    #      waits for its prepass, as files of varying length do) and the first call on the fresh handle (`cold_call_ms`)
    if world == 1 and not dist_on and cert_on:
        # the standing audit of the block certificate: the same analysis once more with every 16th certified
        # (block, channel) pair re-analysed by the exact reference-order kernels, which compare codes, lattice coefficients and
        # the RAW side with what the certified run stored; a difference fails the call (and this bench)
        enc.set_option("cert_audit", 16)
        step()
        torch.cuda.synchronize()
        aud = enc.last_cert_audit()
        enc.set_option("cert_audit", 0)
        out["block_certificate"]["audit"] = {"every": 16, "pairs_audited_equal": int(aud[0]), "pairs_different": int(aud[1])}
        if aud[1] != 0:
            raise SystemExit(3)
    if world == 1 and not dist_on:
        ex = enc.last_expand()
        out["per_handle_shortcuts"] = {"analyses_served_from_kept_search_tables": int(ex[2]), "searches_launched_on_a_wrong_guess": int(ex[3]),
                                       "chunks_from_device_tables": int(ex[0]), "chunks": int(ex[1])}
        enc.set_option("table_cache", 0)
        reps = max(3, min(args.steps, 10))
        step()
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        for _ in range(reps):
            step()
        torch.cuda.synchronize()
        out["ms_per_step_no_kept_tables"] = round((time.perf_counter() - t1) / reps * 1e3, 3)
        enc.set_option("table_cache", 1)

    # ---- end-to-end .sla encode from host PCM (PCIe + bit-pack included); never `value` -------------------------
    if not args.no_e2e and world == 1 and batch is not None:
        enc2 = sla_amd.Encoder(*cap)
        enc2.set_wave_format(nch, bits, rate)
        enc2.set_encode_parameter(order, ltm, lms, ms, win, maxb)
        outs = [np.zeros(4 * nch * int(l) + 65536, np.uint8) for l in batch["lens"]]
        got = enc2.encode_batch(batch["clips"], outs=outs)
        reps = 3 if primary else 2
        t1 = time.perf_counter()
        for _ in range(reps):
            got = enc2.encode_batch(batch["clips"], outs=outs)
        e2e = (time.perf_counter() - t1) / reps
        sla_bytes = int(sum(len(d) for _, d in got))
        h2d = n_own * nch * upload_bytes_per_sample(bits)
        # the same batch in one piece (option batch_lanes = 1: upload, analysis, pack, download one after the other): the bytes
        # must be the same, file by file
        lanes_bytes = [bytes(d) for _, d in got]
        enc2.set_option("batch_lanes", 1)
        plain = enc2.encode_batch(batch["clips"], outs=outs)
        t1 = time.perf_counter()
        for _ in range(reps):
            plain = enc2.encode_batch(batch["clips"], outs=outs)
        e2e_plain = (time.perf_counter() - t1) / reps
        lanes_equal = (lanes_bytes == [bytes(d) for _, d in plain])
        enc2.set_option("batch_lanes", 4)
        if not lanes_equal:
            print(json.dumps({"error": "sla_hip_encode_batch on worker lanes and in one piece differ"}), flush=True)
            sys.exit(3)
        out["end_to_end"] = {"msamples_s": round(n_own * nch / e2e / 1e6, 3), "samples": n_own * nch,
                             "one_piece_msamples_s": round(n_own * nch / e2e_plain / 1e6, 3), "lanes_equal_one_piece": lanes_equal,
                             "sla_bytes": sla_bytes, "files_ok": int(sum(1 for rc, _ in got if rc == 0)),
                             "pcie_bytes_up": h2d, "pcie_bytes_down": sla_bytes,
                             "pcie_frac_up": round(h2d / e2e / 1e9 / PCIE_PEAK_GBS, 4), "pcie_frac_down": round(sla_bytes / e2e / 1e9 / PCIE_PEAK_GBS, 4),
                             "note": "sla_hip_encode_batch: pageable host PCM of every clip -> its own .sla bytes incl. PCIe both ways; "
                                     "pcie_frac_up / _down = bytes that cross the bus in that direction / call time / 63 GB/s (the link is full duplex)"}
        # the drop-in call, clip by clip (what the reference CLI does per file)
        one = batch["clips"][0]
        buf = np.zeros(4 * nch * one.shape[1] + 65536, np.uint8)
        enc2.encode_whole(one, out=buf)
        t1 = time.perf_counter()
        for k in range(20):
            enc2.encode_whole(batch["clips"][k % len(batch["clips"])], out=buf)
        per_clip = (time.perf_counter() - t1) / 20
        out["end_to_end"]["one_clip_per_call_msamples_s"] = round(one.shape[0] * one.shape[1] / per_clip / 1e6, 3)
        out["end_to_end"]["one_clip_per_call_ms"] = round(per_clip * 1e3, 3)
        enc2.close()
    if not args.no_e2e and world == 1 and batch is None:
        if host_pcm is None:
            host_pcm = np.ascontiguousarray(d_pcm[:, :n_file].cpu().numpy())
        enc2 = sla_amd.Encoder(*cap)
        enc2.set_wave_format(nch, bits, rate)
        enc2.set_encode_parameter(order, ltm, lms, ms, win, maxb)
        out_cap = min(4 * nch * n_file + 65536, 0xFFFFFFF0)                # the API's sizes are 32-bit
        outbuf = np.zeros(out_cap, np.uint8)
        reps = 3 if primary else 2

        def timed(src, dst):
            got = enc2.encode_whole(src, out=dst)
            t1 = time.perf_counter()
            for _ in range(reps):
                got = enc2.encode_whole(src, out=dst)
            return got, (time.perf_counter() - t1) / reps

        # files of >= 2 x 32 Mi samples take the streamed path (pieces on worker lanes: upload, kernels and download of
        # different pieces overlap); shorter ones the plain path.  Both are timed; the bytes must be the same.
        data, e2e = timed(host_pcm, outbuf)
        size = len(data)
        enc2.set_option("stream", 0)
        plain, e2e_plain = timed(host_pcm, np.zeros(size + 65536, np.uint8))
        same = bool(len(plain) == size and np.array_equal(plain, data))
        del plain
        enc2.set_option("stream", 1)
        h2d = n_file * nch * upload_bytes_per_sample(bits)
        out["end_to_end"] = {"msamples_s": round(n_file * nch / e2e / 1e6, 3), "samples": n_file * nch,
                             "plain_path_msamples_s": round(n_file * nch / e2e_plain / 1e6, 3), "streamed_equals_plain": same,
                             "sla_bytes": size, "pcie_bytes_up": h2d, "pcie_bytes_down": size,
                             "pcie_frac_up": round(h2d / e2e / 1e9 / PCIE_PEAK_GBS, 4), "pcie_frac_down": round(size / e2e / 1e9 / PCIE_PEAK_GBS, 4),
                             "note": "SLAEncoder_EncodeWhole: pageable host PCM -> .sla bytes incl. PCIe both ways and the device bit-pack; "
                                     "pcie_frac_up / _down = bytes that cross the bus in that direction / call time / 63 GB/s (full duplex: each direction has its own 63 GB/s)"}
        # the same call on page-locked caller memory (hipHostMalloc / hipHostRegister, here torch pinned tensors): DMA without the staging copy
        pin_in = torch.from_numpy(host_pcm).pin_memory()
        pin_out = torch.zeros(size + 65536, dtype=torch.uint8).pin_memory()
        data_p, e2p = timed(pin_in.numpy(), pin_out.numpy())
        out["end_to_end"]["pinned_msamples_s"] = round(n_file * nch / e2p / 1e6, 3)
        out["end_to_end"]["pinned_pcie_frac_up"] = round(n_file * nch * 4 / e2p / 1e9 / PCIE_PEAK_GBS, 4)      # page-locked planes cross as they are: 4 B per sample
        out["end_to_end"]["pinned_pcie_frac_down"] = round(size / e2p / 1e9 / PCIE_PEAK_GBS, 4)
        out["end_to_end"]["pinned_identical"] = bool(len(data_p) == size and np.array_equal(data_p, data))
        enc2.set_option("stream", 0)
        _, e2p_plain = timed(pin_in.numpy(), pin_out.numpy())
        out["end_to_end"]["pinned_plain_path_msamples_s"] = round(n_file * nch / e2p_plain / 1e6, 3)
        del pin_in, pin_out
        enc2.close()
        if not (same and out["end_to_end"]["pinned_identical"]):
            print(json.dumps({"error": "end-to-end paths disagree", "config": cfg}), file=sys.stderr, flush=True)
            raise SystemExit(3)
        if primary and lms in (4, 8, 16, 32) and maxb <= 16384:
            dec = sla_amd.Decoder(nch, maxb, order, ltm, lms)
            stream = bytes(data)
            rc, back = dec.decode_whole(stream, n_file)
            t1 = time.perf_counter()
            for _ in range(reps):
                rc, back = dec.decode_whole(stream, n_file)
            d2e = (time.perf_counter() - t1) / reps
            tm = dec.last_timing()
            out["decode"] = {"msamples_s": round(n_file * nch / d2e / 1e6, 3),
                             "kernels_msamples_s": round(n_file * nch / (tm[2] * 1e3), 3) if tm[2] > 0 else None,
                             "round_trip_identical": bool(rc == 0 and np.array_equal(back, host_pcm)),
                             "note": "SLADecoder_DecodeWhole of the bytes above: .sla in host memory -> planar PCM in host "
                                     "memory incl. PCIe both ways; kernels = CRC16, entropy decode, LMS / long-term / "
                                     "lattice synthesis, de-emphasis between stream events"}
            dec.close()

    # ---- CPU baseline of this configuration (measured before the GPU was touched, see main) ----------------------
    if cpu_results is not None and cfg in cpu_results:
        out["cpu_baseline"] = cpu_results[cfg]
        cpu1 = out["cpu_baseline"]["value"]
        out["speedup_vs_cpu"] = {
            "end_to_end_vs_cpu_full_encode": round(out["end_to_end"]["msamples_s"] / cpu1, 1) if "end_to_end" in out else None,
            "hot_path_vs_cpu_full_encode": round(value / cpu1, 1),
            "note": "like for like is end_to_end (host PCM -> .sla bytes) / the CPU's EncodeWhole; the hot-path figure divides the "
                    "device-resident analysis (no PCIe, no bit-pack) by the CPU's FULL encode and is an upper bound, not a comparison"}
    enc.close()
    del d_pcm, d_lat, d_fin, gathered
    torch.cuda.empty_cache()
    return out


def self_launch(ngpus):
    """start the N ranks of this very command line as a child process: python -m torch.distributed.run ... bench.py ...
    Returns the child's exit code; its stdout (rank 0's JSON line) and stderr pass through."""
    import socket
    import subprocess
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(ngpus),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    return subprocess.run(cmd, env=env).returncode


def main():
    if len(sys.argv) >= 6 and sys.argv[1] == "--cpu-worker":          # a child of cpu_baseline's all-cores leg: never touches the GPU
        print(json.dumps(_cpu_worker((sys.argv[2], int(sys.argv[3]), int(sys.argv[4]), float(sys.argv[5])))), flush=True)
        return
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--config", default="C3", choices=sorted(CONFIGS))
    ap.add_argument("--seconds", type=float, default=None, help="override the audio duration of the primary configuration")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-e2e", action="store_true")
    ap.add_argument("--no-other-configs", action="store_true", help="N = 1: skip the other three configurations' lines of `other_configs`")
    ap.add_argument("--cpu-seconds", type=float, default=15.0, help="single-thread CPU work of the primary configuration's baseline (a third of it for every other leg)")
    ap.add_argument("--sync-gather", action="store_true", help="N > 1: wait for each step's all-gather before the next step starts")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend (nccl = RCCL; gloo only to rehearse on one GPU)")
    ap.add_argument("--clips", type=int, default=125, help="C4, weak scaling: clips per GPU and step (1000 clips over 8 GPUs)")
    ap.add_argument("--total-clips", type=int, default=1000, help="C4, strong scaling: clips per step over all GPUs")
    ap.add_argument("--scaling", default="weak", choices=("weak", "strong"),
                    help="N > 1: weak = one file N times the configuration's length (per-GPU work fixed); strong = the "
                         "configuration's own length (C4: --total-clips) split over the N ranks")
    ap.add_argument("--force-collectives", action="store_true",
                    help="N = 1: still bring up the process group and run the primary configuration through the N > 1 path (scan, count "
                         "exchange, bounds, shard analyse, all-gather of the residual planes), so that the RCCL calls execute on a one-GPU box")
    ap.add_argument("--launch-check", action="store_true",
                    help="only bring the ranks up (process group of --backend, no GPU work) and print the line's launch fields")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # `python bench.py --gpus N` as the driver may run it: this process has not touched the GPU (no torch import yet)
        # and never will -- it starts the N ranks as a CHILD process, relays what rank 0 prints and exits with its code
        raise SystemExit(self_launch(args.gpus))

    import slalibs as S

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        print(json.dumps({"error": "--gpus %d but WORLD_SIZE is %d: launch with --nproc-per-node %d (or without torchrun)"
                                   % (args.gpus, world, args.gpus)}), file=sys.stderr, flush=True)
        raise SystemExit(2)
    if args.launch_check:
        import torch.distributed as dist
        seen = 1
        if world > 1:
            dist.init_process_group("gloo" if args.backend != "nccl" else args.backend)
            seen = dist.get_world_size()
            dist.barrier()
        if rank == 0:
            print(json.dumps({"launch_check": True, "n_gpus": world, "ranks_seen": seen, "scaling": args.scaling}), flush=True)
        if world > 1:
            dist.destroy_process_group()
        return
    want_others = (world == 1 and not args.no_other_configs and args.seconds is None)
    # ---- CPU legs first, while this process has not touched the GPU (they start worker processes) ---------------
    cpu_results = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        cpu_results = {args.config: cpu_baseline(S, args.config, args.cpu_seconds, args.cpu_seconds / 3)}
        if want_others:
            for cfg in sorted(CONFIGS):
                if cfg != args.config:
                    cpu_results[cfg] = cpu_baseline(S, cfg, args.cpu_seconds / 3, 0.0)

    import torch
    import sla_amd
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the HIP path has no CPU fallback")
    dev_index = local_rank % max(torch.cuda.device_count(), 1)
    torch.cuda.set_device(dev_index)
    dist = None
    dist_on = (world > 1) or args.force_collectives
    if dist_on:
        import torch.distributed as dist
        if world == 1:                    # --force-collectives without a launcher: a one-rank rendezvous of our own
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", str(29500 + os.getpid() % 2000))
            os.environ.setdefault("RANK", "0")
            os.environ.setdefault("WORLD_SIZE", "1")
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", dev_index))
        else:
            dist.init_process_group(args.backend)

    ranks_seen = dist.get_world_size() if dist_on else 1
    res = run_config(torch, sla_amd, S, args.config, args, rank, world, True, cpu_results)
    if rank == 0:
        out = {"metric": "encode Msamples/s (LPC+residual path), verified bit-exact vs the oracle in this run",
               "value": res.pop("value"), "unit": res.pop("unit"), "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
               "ms_per_step": res.pop("ms_per_step"), "higher_is_better": True, "scaling": args.scaling, "vs_baseline": None,
               "dtype": "f64+int32", "data": "synthetic",
               ("nccl_ranks_seen" if args.backend == "nccl" else "%s_ranks_seen" % args.backend): ranks_seen}
        # {config: [Msamples/s, ms per step, roofline.frac of its dominant kernel, 1-core CPU Msamples/s]}: filled in below, kept
        # among the first keys (and repeated at the very end) so that no truncation of this long line loses a configuration
        summary = {args.config: [out["value"], out["ms_per_step"], res["roofline"]["frac"],
                                 (res.get("cpu_baseline") or {}).get("value")]}
        out["summary"] = summary
        out.update(res)
        if args.force_collectives:
            out["collectives_forced"] = ("world %d: process group of backend %s, exchange_counts, shard bounds, shard_analyze and the (overlapped) "
                                         "all-gather of the residual planes run exactly as for N > 1" % (world, args.backend))
        out["device"] = sla_amd.device_name()
        if want_others:
            others = {}
            for cfg in sorted(CONFIGS):
                if cfg == args.config:
                    continue
                sub = argparse.Namespace(**vars(args))
                sub.steps, sub.warmup = max(3, min(args.steps, 6)), 1
                r = run_config(torch, sla_amd, S, cfg, sub, rank, world, False, cpu_results)
                r["steps"], r["warmup"] = sub.steps, sub.warmup
                others[cfg] = r
                summary[cfg] = [r["value"], r["ms_per_step"], r["roofline"]["frac"], (r.get("cpu_baseline") or {}).get("value")]
            out["other_configs"] = others
        out["summary_again"] = dict(summary)
        print(json.dumps(out), flush=True)
    if dist_on:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
