"""SLAEncoder_EncodeWhole of one long bench file from pageable / page-locked memory, default options (dev tool):
python tests/tools/e2e_quick.py CFG [reps] [option=value ...]"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
import torch          # noqa: E402
import bench          # noqa: E402
import sla_amd        # noqa: E402

cfg = sys.argv[1]
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 4
nch, bits, rate, seconds, order, ltm, lms, ms, win, maxb, cap = bench.CONFIGS[cfg]
n = rate * seconds
pcm_dev = bench.synth_device(torch, nch, n, bits, rate, 0, n)
pcm = pcm_dev.cpu().numpy()
pinned = torch.empty((nch, n), dtype=torch.int32).pin_memory()
pinned.copy_(pcm_dev.cpu())
del pcm_dev
cap_bytes = min(4 * nch * n + (1 << 20), 0xFFFFFFF0)
out = np.zeros(cap_bytes, np.uint8)
pin_out = torch.zeros(cap_bytes, dtype=torch.uint8).pin_memory()
enc = sla_amd.Encoder(*cap)
enc.set_wave_format(nch, bits, rate)
enc.set_encode_parameter(order, ltm, lms, ms, win, maxb)
for a in sys.argv[3:]:
    k, v = a.split("=")
    enc.set_option(k, float(v))
for src, dst, name in ((pcm, out, "pageable"), (pinned.numpy(), pin_out.numpy(), "page-locked")):
    ts = []
    for _ in range(reps + 1):
        t0 = time.perf_counter()
        got = enc.encode_whole(src, out=dst)
        ts.append((time.perf_counter() - t0) * 1e3)
    print("%s %s GPU_MAX_HW_QUEUES=%s: %s ms; best %.0f Msamples/s" % (cfg, name, os.environ.get("GPU_MAX_HW_QUEUES", "-"), " ".join("%.1f" % t for t in ts[1:]), n * nch / min(ts[1:]) / 1e3), flush=True)
