"""SLAEncoder_EncodeWhole on one file from pageable / page-locked host memory for several stream_piece sizes (dev tool):
python tests/tools/stream_piece_sweep.py CFG [SECONDS]"""
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
out = np.zeros(4 * nch * n + (1 << 20), np.uint8)
ref = None
for piece in (0, 2 << 20, 4 << 20, 8 << 20, 16 << 20, 32 << 20):
    enc = sla_amd.Encoder(*cap)
    enc.set_wave_format(nch, bits, rate)
    enc.set_encode_parameter(order, ltm, lms, ms, win, maxb)
    if piece == 0:
        enc.set_option("stream", 0)
    else:
        enc.set_option("stream_piece", piece)
    res = []
    for src, name in ((pcm, "pageable"), (pinned.numpy(), "page-locked")):
        for _ in range(2):
            got = enc.encode_whole(src, out=out)
        t0 = time.perf_counter()
        reps = 5
        for _ in range(reps):
            got = enc.encode_whole(src, out=out)
        dt = (time.perf_counter() - t0) / reps
        h = hash(bytes(got)) if not isinstance(got, (bytes, bytearray)) else hash(bytes(got))
        if ref is None:
            ref = h
        res.append("%s %.2f ms = %.0f Msamples/s%s" % (name, dt * 1e3, n * nch / dt / 1e6, "" if h == ref else "  !! bytes differ"))
    print("%s %ds piece %s: %s" % (cfg, seconds, "plain" if piece == 0 else "%d Mi" % (piece >> 20), "; ".join(res)), flush=True)
    enc.close()
