"""Decode throughput of SLADecoder_DecodeWhole on a BASELINE-shaped stream (GPU box).
usage: python tests/tools/bench_decode.py [C2|C3|C5] [seconds]"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import numpy as np
import torch
torch.cuda.init()
import sla_amd
import slalibs as S

CFG = {"C2": (1, 16, 48000, 16, 1, 8, 0, 4096, 600), "C3": (2, 24, 48000, 32, 3, 8, 1, 4096, 600),
       "C5": (8, 24, 96000, 48, 3, 8, 0, 8192, 120)}
name = sys.argv[1] if len(sys.argv) > 1 else "C2"
nch, bits, rate, order, ltm, lms, ms, mb, secs = CFG[name]
if len(sys.argv) > 2:
    secs = int(sys.argv[2])
n = rate * secs
pcm = S.synth_pcm(nch, n, bits, rate)
enc = sla_amd.Encoder(nch, mb, order, ltm, lms)
enc.set_wave_format(nch, bits, rate)
enc.set_encode_parameter(order, ltm, lms, ms, 1, mb)
data = enc.encode_whole(pcm)
enc.close()
print(name, "samples/ch", n, "sla bytes", len(data))
dec = sla_amd.Decoder(nch, mb, order, ltm, lms)
for it in range(4):
    t0 = time.time()
    rc, got = dec.decode_whole(data, n)
    dt = time.time() - t0
    t = dec.last_timing()
    print("rc", rc, "wall %.1f ms" % (dt * 1e3), "upload %.2f walk %.2f kernels %.2f download %.2f total %.2f batches %d" % tuple(t),
          "-> %.0f Msamples/s end to end, %.0f Msamples/s kernels" % (n * nch / t[4] / 1e3, n * nch / t[2] / 1e3))
print("identical", bool(np.array_equal(got, pcm)))
