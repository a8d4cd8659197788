"""timeline of SLAEncoder_EncodeWhole on one 10-second 48 kHz 16-bit stereo clip (dev tool): SLA_HIP_TRACE=1 python clip_trace.py"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import sla_amd
import slalibs as S
pcm = S.synth_pcm(2, 480000, 16, 48000, seed=4001)
enc = sla_amd.Encoder(2, 4096, 16, 1, 8)
enc.set_wave_format(2, 16, 48000)
enc.set_encode_parameter(16, 1, 8, 1, 1, 4096)
buf = np.zeros(4 * 2 * 480000 + 65536, np.uint8)
for _ in range(3):
    enc.encode_whole(pcm, out=buf)
os.environ["X"] = "1"
t = time.perf_counter()
for _ in range(20):
    enc.encode_whole(pcm, out=buf)
dt = (time.perf_counter() - t) / 20
print("per clip %.3f ms = %.1f Msamples/s; timing %s" % (dt * 1e3, 960000 / dt / 1e6, [round(x, 3) for x in enc.last_timing()]))
