"""per-step wall times of the hot path on one file (dev tool): python tests/tools/step_jitter.py CFG [SECONDS] [STEPS]
prints the distribution and, for every step slower than 1.3 x the median, its stage times"""
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
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 300
n = rate * seconds
stride = (n + 63) // 64 * 64
d_pcm = torch.zeros((nch, stride), dtype=torch.int32, device="cuda")
d_pcm[:, :n] = bench.synth_device(torch, nch, n, bits, rate, 0, n)
d_lat = torch.zeros_like(d_pcm)
d_fin = torch.zeros_like(d_pcm)
enc = sla_amd.Encoder(*cap)
enc.set_wave_format(nch, bits, rate)
enc.set_encode_parameter(order, ltm, lms, ms, win, maxb)
for kv in sys.argv[4:]:
    k, v = kv.split("=")
    enc.set_option(k, float(v))
enc.bind_residual_planes(d_lat.data_ptr(), d_fin.data_ptr(), stride)
torch.cuda.synchronize()
for _ in range(3):
    enc.analyze_device(d_pcm.data_ptr(), stride, n)
if os.environ.get("SLA_JITTER_NOGC"):
    import gc
    gc.collect()
    gc.disable()
t = np.zeros(steps)
parts = []


def cg():
    try:
        d = dict(l.split() for l in open("/sys/fs/cgroup/cpu.stat").read().strip().splitlines())
        return int(d["usage_usec"]), int(d.get("nr_throttled", 0)), int(d.get("throttled_usec", 0))
    except OSError:
        return 0, 0, 0


cg0, pt0, w0 = cg(), os.times(), time.perf_counter()
for i in range(steps):
    t0 = time.perf_counter()
    parts.append(enc.analyze_device(d_pcm.data_ptr(), stride, n))
    t[i] = (time.perf_counter() - t0) * 1e3
cg1, pt1, w1 = cg(), os.times(), time.perf_counter()
wall = w1 - w0
print("wall %.3f s: this process %.2f cores (user %.2f sys %.2f), cgroup %.2f cores, throttled %d times for %.1f ms"
      % (wall, (pt1[0] + pt1[1] - pt0[0] - pt0[1]) / wall, (pt1[0] - pt0[0]) / wall, (pt1[1] - pt0[1]) / wall,
         (cg1[0] - cg0[0]) / 1e6 / wall, cg1[1] - cg0[1], (cg1[2] - cg0[2]) / 1e3))
med = np.median(t)
print("%s %ds: %d steps, median %.3f mean %.3f min %.3f p90 %.3f p99 %.3f max %.3f ms" % (cfg, seconds, steps, med, t.mean(), t.min(), np.percentile(t, 90), np.percentile(t, 99), t.max()))
for i in range(steps):
    if t[i] > 1.3 * med:
        p = parts[i]
        print("  step %3d: %.3f ms | prepass %.2f search %.2f blocks %.2f lattice %.2f tail %.2f acf %.2f host %.2f" % (i, t[i], p[0], p[1], p[2], p[3], p[4], p[8], p[5]))
