"""sla_hip_encode_batch of the C4 batch (125 ten-second 16-bit stereo clips, pageable host memory) for several lane counts (dev tool):
[SLA_HIP_TRACE=1] python tests/tools/batch_lanes_sweep.py [clips]"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import torch          # noqa: E402
import bench          # noqa: E402
import slalibs as S   # noqa: E402
import sla_amd        # noqa: E402

torch.cuda.init()
nch, bits, rate, seconds, order, ltm, lms, ms, win, maxb, cap = bench.CONFIGS["C4"]
nclips = int(sys.argv[1]) if len(sys.argv) > 1 else 125
n = rate * seconds
distinct = [S.synth_pcm(nch, n, bits, rate, seed=4000 + k) for k in range(16)]
clips = [distinct[k % 16].copy() for k in range(nclips)]          # every clip its own host memory
outs = [np.zeros(4 * nch * n + 65536, np.uint8) for _ in clips]
ref = None
for lanes in (1, 2, 3, 4, 6, 4, 1):
    enc = sla_amd.Encoder(*cap)
    enc.set_wave_format(nch, bits, rate)
    enc.set_encode_parameter(order, ltm, lms, ms, win, maxb)
    enc.set_option("batch_lanes", lanes)
    for _ in range(2):
        got = enc.encode_batch(clips, outs=outs)
    ts = []
    for _ in range(4):
        t0 = time.perf_counter()
        got = enc.encode_batch(clips, outs=outs)
        ts.append((time.perf_counter() - t0) * 1e3)
    h = [hash(bytes(d)) for _, d in got]
    if ref is None:
        ref = h
    print("lanes %d: %s ms; best %.0f Msamples/s%s" % (lanes, " ".join("%.1f" % t for t in ts), nclips * n * nch / min(ts) / 1e3, "" if h == ref else "  !! bytes differ"), flush=True)
    enc.close()
