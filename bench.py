#!/usr/bin/env python3
"""bench.py -- encode Msamples/s of the SLA LPC+residual hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W [--config C2|C3|C4|C5] [--seconds S]

One "step" = one pass of the hot path (sla_hip_analyze_device: prepass -> partition-search LPC ->
block LPC + quantiser -> PARCOR lattice -> long-term + LMS + Rice parameter) over one batch of
synthetic PCM that is already resident in HBM when the timed region starts.  N > 1: launched by
torch.distributed.run, one rank per GPU, every rank encodes its own shard of the (N x larger)
job (frames are independent -> weak scaling) and the residual stream is re-assembled by one RCCL
all-gather per step, as the north star prescribes.

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
ALGO_BYTES_PER_SAMPLE = 8      # 4 B int32 PCM read + 4 B int32 final residual written


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--config", default="C2", choices=sorted(CONFIGS))
    ap.add_argument("--seconds", type=float, default=None, help="override the audio duration")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-e2e", action="store_true")
    ap.add_argument("--sync-gather", action="store_true", help="N > 1: wait for each step's all-gather before the next step starts")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend (nccl = RCCL; gloo only to rehearse on one GPU)")
    ap.add_argument("--clips", type=int, default=125, help="C4 only: clips per GPU and step (1000 clips over 8 GPUs)")
    args = ap.parse_args()

    import torch
    import sla_amd
    import slalibs as S

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the HIP path has no CPU fallback")
    dev_index = local_rank % max(torch.cuda.device_count(), 1)
    torch.cuda.set_device(dev_index)
    dist = None
    if world > 1:
        import torch.distributed as dist
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", dev_index))
        else:
            dist.init_process_group(args.backend)

    nch, bits, rate, seconds, order, ltm, lms, ms, win, maxb, cap = CONFIGS[args.config]
    if args.seconds is not None:
        seconds = args.seconds
    n = int(rate * seconds)
    batch = None
    if args.config == "C4":
        # a batch of clips per step (BASELINE config 3: 1000 clips over 8 GPUs), laid out back to back on
        # 1024-sample boundaries and analysed in ONE pipeline pass (sla_hip_analyze_batch_device)
        clip_n, tile = n, 1024
        pitch = (clip_n + tile - 1) // tile * tile
        distinct = [S.synth_pcm(nch, clip_n, bits, rate, seed=4000 + 16 * rank + k) for k in range(min(16, args.clips))]
        clips = [distinct[k % len(distinct)] for k in range(args.clips)]
        batch = {"starts": np.arange(args.clips, dtype=np.uint32) * pitch, "lens": np.full(args.clips, clip_n, np.uint32),
                 "clips": clips, "span": args.clips * pitch}
        n = clip_n * args.clips                          # samples per channel that are audio
        stride = batch["span"]
        d_pcm = torch.zeros((nch, stride), dtype=torch.int32, device="cuda")
        for k, c in enumerate(clips):
            d_pcm[:, k * pitch:k * pitch + clip_n] = torch.from_numpy(c).cuda()
        pcm = None
    else:
        # every rank encodes its own, different shard (seed by rank)
        pcm = S.synth_pcm(nch, n, bits, rate, seed=12345 + rank)
        stride = (n + 63) // 64 * 64
        d_pcm = torch.zeros((nch, stride), dtype=torch.int32, device="cuda")
        d_pcm[:, :n] = torch.from_numpy(pcm).cuda()
    # N > 1 over RCCL: the all-gather of one step's residual planes travels while the next step is analysed into
    # a second set of planes (two sets take turns), so the collective over xGMI and the kernels overlap
    overlap = (world > 1 and args.backend == "nccl" and not args.sync_gather)
    nbuf = 2 if overlap else 1
    d_lat = [torch.zeros((nch, stride), dtype=torch.int32, device="cuda") for _ in range(nbuf)]
    d_fin = [torch.zeros((nch, stride), dtype=torch.int32, device="cuda") for _ in range(nbuf)]
    gathered = [torch.empty((world, nch, stride), dtype=torch.int32, device="cuda") for _ in range(nbuf)] if world > 1 else None
    works = [None] * nbuf

    enc = sla_amd.Encoder(*cap)
    enc.set_wave_format(nch, bits, rate)
    enc.set_encode_parameter(order, ltm, lms, ms, win, maxb)
    enc.bind_residual_planes(d_lat[0].data_ptr(), d_fin[0].data_ptr(), stride)
    torch.cuda.synchronize()

    from sla_amd import dist as sdist
    step_no = [0]
    span_ms = np.zeros(4)       # on-device execution time of k_lpc_blocks, k_lattice, k_ltm_acf, k_tail (summed over steps)

    def settle(b):
        if works[b] is not None:
            works[b].wait()
            torch.cuda.current_stream().synchronize()
            works[b] = None

    def step():
        b = step_no[0] % nbuf
        step_no[0] += 1
        settle(b)                                             # the planes of two steps ago have been gathered
        if nbuf > 1:
            enc.bind_residual_planes(d_lat[b].data_ptr(), d_fin[b].data_ptr(), stride)
        if batch is not None:
            t, _ = enc.analyze_batch_device(d_pcm.data_ptr(), stride, batch["span"], batch["starts"], batch["lens"])
        else:
            t = enc.analyze_device(d_pcm.data_ptr(), stride, n)
        span_ms[:] += np.array(enc.last_kernel_ms())
        if world > 1:
            if overlap:
                works[b] = sdist.all_gather_planes(d_fin[b], gathered[b], async_op=True)[1]
            elif args.backend == "nccl":
                sdist.all_gather_planes(d_fin[b], gathered[b])   # RCCL over xGMI: re-assemble the residual stream
            else:                                             # rehearsal backend: stage through the host
                gathered[b].copy_(sdist.all_gather_planes(d_fin[b].cpu()))
        return t

    for _ in range(args.warmup):
        step()
    for b in range(nbuf):
        settle(b)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    kernel_ms = np.zeros(12)
    span_ms[:] = 0.0
    t0 = time.perf_counter()
    for _ in range(args.steps):
        kernel_ms += np.array(step())
    for b in range(nbuf):
        settle(b)                                             # every collective of the timed steps has landed
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    if world > 1:
        elapsed = sdist.max_over_ranks(elapsed, "cuda" if args.backend == "nccl" else "cpu")
    kernel_ms /= max(args.steps, 1)
    span_ms /= max(args.steps, 1)

    total_samples = float(n) * nch * world * args.steps
    value = total_samples / elapsed / 1e6

    out = {
        "metric": "encode Msamples/s (LPC+residual path), bit-exact vs oracle",
        "value": round(value, 3), "unit": "Msamples/s", "n_gpus": world, "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": round(elapsed / max(args.steps, 1) * 1e3, 3),
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f64+int32",
        "data": "synthetic",
        "config": {"workload": WORKLOAD_NAME[args.config] + (", batch of %d clips per GPU in one pass" % args.clips if batch else "")
                               + (" x%d ranks" % world if world > 1 else ""),
                   "name": args.config, "channels": nch, "bits": bits, "rate": rate, "seconds": seconds,
                   "parcor_order": order, "longterm_order": ltm, "lms_order": lms,
                   "max_block_samples": maxb, "samples_per_step_per_gpu": n * nch,
                   "parallelism": "frames sharded over %d GPU(s), RCCL all-gather of residuals%s" % (world, " overlapped with the next step" if overlap else "")},
    }

    if rank == 0:
        # ---- roofline of the dominant kernel (HIP-event durations measured inside the library, on the
        #      stream the kernels run on) ---------------------------------------------------------------
        # per-step kernel time, launches per step and passes over the data, by kernel name (k_lpc runs twice
        # over the file: partition search and chosen blocks; chunked stages launch once per chunk)
        nchunks = max(int(round(kernel_ms[9])), 1)
        exact_search = kernel_ms[11] > 0.5      # partition search ran on tile sums, k_lpc only on the chosen blocks
        # the four big kernels report their own execution time (first wave in to last wave out, constant-rate device
        # clock = what rocprofv3 --kernel-trace shows); stream-event pairs also count the time a launch waits behind
        # kernels of the other streams, so they are kept for the stages that have nothing else
        ev = {"k_lpc_blocks": kernel_ms[2], "k_lattice": kernel_ms[3], "k_ltm_acf": kernel_ms[8], "k_tail": kernel_ms[4]}
        dev = dict(zip(("k_lpc_blocks", "k_lattice", "k_ltm_acf", "k_tail"), span_ms))
        kernels = {"k_prepass": (kernel_ms[0], 1, 1)}
        for name in ev:
            kernels[name] = (ev[name], nchunks, 1)                   # HIP events on the kernel's own stream, as the contract asks
        if exact_search and kernel_ms[10] > 0:
            kernels["k_lpc"] = (kernel_ms[1], nchunks, 1)            # tile sums + rerun of the flagged windows as serial chains: the chains dominate
        elif exact_search:
            kernels["k_acf_tiles"] = (kernel_ms[1], nchunks, 1)      # the event pair also spans k_search_finish and k_plan
        else:
            kernels["k_lpc"] = (kernel_ms[1], nchunks, 1)            # serial-chain partition search
        dom = max(kernels, key=lambda k: kernels[k][0])
        t_step, launches, passes = kernels[dom]
        launch_ms = t_step / launches
        algo_bytes = float(n) * nch * ALGO_BYTES_PER_SAMPLE * passes / launches      # per launch
        achieved = algo_bytes / (launch_ms * 1e-3) / 1e9 if launch_ms > 0 else 0.0
        out["roofline"] = {"bound": "hbm", "kernel": dom, "achieved": round(achieved, 2),
                           "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 5),
                           "traffic": None, "kernel_ms": round(float(launch_ms), 4), "launches_per_step": launches,
                           "algorithmic_bytes_per_launch": algo_bytes,
                           "kernel_ms_running": round(float(dev[dom] / launches), 4) if dom in dev and dev[dom] > 0 else None,
                           "note": "kernel_ms = average duration of one launch of this kernel inside the timed region, between HIP events "
                                   "on the stream it is launched on (agrees with rocprofv3 --kernel-trace --stats of the same command, "
                                   "profiles/r1_kernel_stats_c2.csv; both include the time a launch shares the SIMDs with, or waits "
                                   "behind, the next chunk's kernels on the other streams); kernel_ms_running = first workgroup in to "
                                   "last workgroup out on the device's 100 MHz clock, i.e. without that wait"}
        # HBM traffic per launch of that kernel from the committed rocprofv3 --pmc passes (same config only)
        try:
            pmc = json.load(open(os.path.join(ROOT, "profiles", "pmc_traffic_%s.json" % args.config.lower())))
            if args.seconds is None:
                out["roofline"]["traffic"] = pmc["bytes_per_launch"][dom]["total"]
                out["roofline"]["traffic_source"] = pmc["source"]
        except (OSError, KeyError, ValueError):
            pass
        out["kernel_ms_on_device"] = {k: round(float(v), 4) for k, v in dev.items()}
        out["stage_ms"] = {"k_prepass": round(float(kernel_ms[0]), 4),
                           ("search_tile_sums" if exact_search else "k_lpc_search"): round(float(kernel_ms[1]), 4),
                           "search_fallback_groups": round(float(kernel_ms[10]), 2),
                           "k_lpc_blocks": round(float(kernel_ms[2]), 4), "k_lattice": round(float(kernel_ms[3]), 4),
                           "k_tail": round(float(kernel_ms[4]), 4), "k_ltm_acf": round(float(kernel_ms[8]), 4), "host_plan": round(float(kernel_ms[5]), 4),
                           "host_longterm": round(float(kernel_ms[6]), 4), "analyze_total": round(float(kernel_ms[7]), 4)}
        out["device"] = sla_amd.device_name()

        # ---- end-to-end .sla encode from host PCM (PCIe + host bit-pack included); never `value` ------
        if not args.no_e2e and world == 1 and batch is not None:
            enc2 = sla_amd.Encoder(*cap)
            enc2.set_wave_format(nch, bits, rate)
            enc2.set_encode_parameter(order, ltm, lms, ms, win, maxb)
            outs = [np.zeros(4 * nch * int(l) + 65536, np.uint8) for l in batch["lens"]]
            got = enc2.encode_batch(batch["clips"], outs=outs)
            reps = 3
            t1 = time.perf_counter()
            for _ in range(reps):
                got = enc2.encode_batch(batch["clips"], outs=outs)
            e2e = (time.perf_counter() - t1) / reps
            out["end_to_end"] = {"msamples_s": round(n * nch / e2e / 1e6, 3), "samples": n * nch,
                                 "sla_bytes": int(sum(len(d) for _, d in got)), "files_ok": int(sum(1 for rc, _ in got if rc == 0)),
                                 "note": "sla_hip_encode_batch: pageable host PCM of every clip -> its own .sla bytes incl. PCIe both ways"}
            enc2.close()
        if not args.no_e2e and world == 1 and batch is None:
            enc2 = sla_amd.Encoder(*cap)
            enc2.set_wave_format(nch, bits, rate)
            enc2.set_encode_parameter(order, ltm, lms, ms, win, maxb)
            m = n
            sub = pcm
            outbuf = np.zeros(4 * nch * m + 65536, np.uint8)
            enc2.encode_whole(sub, out=outbuf)
            reps = 3
            t1 = time.perf_counter()
            for _ in range(reps):
                data = enc2.encode_whole(sub, out=outbuf)
            e2e = (time.perf_counter() - t1) / reps
            out["end_to_end"] = {"msamples_s": round(m * nch / e2e / 1e6, 3), "samples": m * nch,
                                 "sla_bytes": len(data), "note": "SLAEncoder_EncodeWhole: pageable host PCM -> .sla bytes incl. PCIe both ways and the device bit-pack"}
            enc2.close()
            # ---- and back: SLADecoder_DecodeWhole of those bytes must return the input (round trip at full size)
            if lms in (4, 8, 16, 32) and maxb <= 16384:
                dec = sla_amd.Decoder(nch, maxb, order, ltm, lms)
                stream = bytes(data)
                rc, back = dec.decode_whole(stream, m)
                t1 = time.perf_counter()
                for _ in range(reps):
                    rc, back = dec.decode_whole(stream, m)
                d2e = (time.perf_counter() - t1) / reps
                tm = dec.last_timing()
                out["decode"] = {"msamples_s": round(m * nch / d2e / 1e6, 3),
                                 "kernels_msamples_s": round(m * nch / (tm[2] * 1e3), 3) if tm[2] > 0 else None,
                                 "round_trip_identical": bool(rc == 0 and np.array_equal(back, sub)),
                                 "note": "SLADecoder_DecodeWhole of the bytes above: .sla in host memory -> planar PCM in host "
                                         "memory incl. PCIe both ways; kernels = CRC16, entropy decode, LMS / long-term / "
                                         "lattice synthesis, de-emphasis between stream events"}
                dec.close()

        # ---- CPU baseline on this box's host cores, same workload, bounded sample -------------------
        if not args.no_cpu_baseline and world == 1:
            p = S.make_params(nch, bits, rate, order, ltm, lms, ms, win, maxb, cap=cap)
            ref = S.ref()
            checker, kind = (ref, "reference") if ref is not None else (S.oracle(), "port")
            if batch is not None:                        # clip by clip, like the reference CLI would: the first 24 clips
                k = min(24, len(batch["clips"]))
                m = int(batch["lens"][0]) * k
                t1 = time.perf_counter()
                for c in batch["clips"][:k]:
                    ret, data = checker.encode_whole(p, c)
                    assert ret == 0
                cpu_s = time.perf_counter() - t1
            else:
                m = min(n, int(rate * 600 / nch))            # about 10-30 s of single-thread CPU work
                sub = np.ascontiguousarray(pcm[:, :m])
                t1 = time.perf_counter()
                ret, data = checker.encode_whole(p, sub)
                cpu_s = time.perf_counter() - t1
                assert ret == 0
            out["cpu_baseline"] = {"value": round(m * nch / cpu_s / 1e6, 3), "unit": "Msamples/s", "cores": 1,
                                   "kind": kind, "seconds": round(cpu_s, 2),
                                   "sample": ("first %d clips, one after the other; " % min(24, len(batch["clips"])) if batch is not None else "")
                                             + "first %d samples x %d ch of the same workload, full single-thread "
                                             "EncodeWhole (%s)" % (m, nch, "unmodified reference, oracle/_ref"
                                                                   if kind == "reference" else "oracle restatement")}
            out["speedup_vs_cpu"] = round(value / out["cpu_baseline"]["value"], 1)
        print(json.dumps(out), flush=True)

    enc.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
