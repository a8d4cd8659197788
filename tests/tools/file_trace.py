"""host timeline of sla_hip_analyze_device on one device-resident file of a bench configuration (dev tool):
SLA_HIP_TRACE=1 python tests/tools/file_trace.py C3 [seconds]"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import numpy as np
import torch
import bench as B
import sla_amd
cfg = sys.argv[1] if len(sys.argv) > 1 else "C3"
nch, bits, rate, seconds, order, ltm, lms, ms, win, maxb, cap = B.CONFIGS[cfg]
opts = [a.split("=") for a in sys.argv[2:] if "=" in a]
secs = [a for a in sys.argv[2:] if "=" not in a]
if secs:
    seconds = float(secs[0])
n = int(rate * seconds)
stride = (n + 63) // 64 * 64
d_pcm = torch.zeros((nch, stride), dtype=torch.int32, device="cuda")
d_pcm[:, :n] = B.synth_device(torch, nch, n, bits, rate, 0, n)
enc = sla_amd.Encoder(*cap)
enc.set_wave_format(nch, bits, rate)
enc.set_encode_parameter(order, ltm, lms, ms, win, maxb)
for k, v in opts:
    enc.set_option(k, float(v))
d_lat = torch.zeros((nch, stride), dtype=torch.int32, device="cuda")
d_fin = torch.zeros((nch, stride), dtype=torch.int32, device="cuda")
enc.bind_residual_planes(d_lat.data_ptr(), d_fin.data_ptr(), stride)
torch.cuda.synchronize()
for _ in range(3):
    enc.analyze_device(d_pcm.data_ptr(), stride, n)
print("---- traced steps", file=sys.stderr, flush=True)
t = time.perf_counter()
for _ in range(6):
    tm = enc.analyze_device(d_pcm.data_ptr(), stride, n)
dt = (time.perf_counter() - t) / 6
print("%s %s %.0f s: %.3f ms per step = %.1f Msamples/s; timing %s; expand %s" % (cfg, opts, seconds, dt * 1e3, nch * n / dt / 1e6, [round(x, 3) for x in tm], enc.last_expand()))
