"""SLAEncoder_EncodeWhole of one long file from pageable / page-locked memory for several lane counts and piece sizes (dev tool):
python tests/tools/stream_lanes_sweep.py CFG [SECONDS]"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
import torch          # noqa: E402
import bench          # noqa: E402
import sla_amd        # noqa: E402

cfg = sys.argv[1]
nch, bits, rate, seconds, order, ltm, lms, ms, win, maxb, cap = bench.CONFIGS[cfg]
if len(sys.argv) > 2 and sys.argv[2]:
    seconds = int(sys.argv[2])
n = rate * seconds
pcm_dev = bench.synth_device(torch, nch, n, bits, rate, 0, n)
pcm = pcm_dev.cpu().numpy()
pinned = torch.empty((nch, n), dtype=torch.int32).pin_memory()
pinned.copy_(pcm_dev.cpu())
del pcm_dev
cap_bytes = min(4 * nch * n + (1 << 20), 0xFFFFFFF0)            # the API's sizes are 32-bit
out = np.zeros(cap_bytes, np.uint8)
pin_out = torch.zeros(cap_bytes, dtype=torch.uint8).pin_memory()
ref = None
for lanes, piece in ((4, 32 << 20), (5, 32 << 20), (6, 32 << 20), (6, 24 << 20), (6, 16 << 20), (5, 24 << 20)):
    enc = sla_amd.Encoder(*cap)
    enc.set_wave_format(nch, bits, rate)
    enc.set_encode_parameter(order, ltm, lms, ms, win, maxb)
    enc.set_option("stream_lanes", lanes)
    enc.set_option("stream_piece", piece)
    res = []
    for src, dst, name in ((pcm, out, "pageable"), (pinned.numpy(), pin_out.numpy(), "page-locked")):
        for _ in range(2):
            got = enc.encode_whole(src, out=dst)
        t0 = time.perf_counter()
        reps = 4
        for _ in range(reps):
            got = enc.encode_whole(src, out=dst)
        dt = (time.perf_counter() - t0) / reps
        h = hash(bytes(got))
        if ref is None:
            ref = h
        res.append("%s %.2f ms = %.0f Msamples/s%s" % (name, dt * 1e3, n * nch / dt / 1e6, "" if h == ref else "  !! bytes differ"))
    print("%s %ds lanes %d piece %d Mi: %s" % (cfg, seconds, lanes, piece >> 20, "; ".join(res)), flush=True)
    enc.close()
