#!/usr/bin/env python3
"""where the end-to-end .sla encode time goes (GPU box helper): python tests/e2e_breakdown.py [seconds]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import sla_amd, slalibs as S

secs = float(sys.argv[1]) if len(sys.argv) > 1 else 600
n = int(48000 * secs)
pcm = S.synth_pcm(1, n, 16, 48000)
torch.cuda.init()
enc = sla_amd.Encoder(1, 4096, 16, 1, 8)
enc.set_wave_format(1, 16, 48000)
enc.set_encode_parameter(16, 1, 8, 0, 1, 4096)
stride = (n + 63) // 64 * 64
d = torch.zeros((1, stride), dtype=torch.int32, device="cuda")
d[:, :n] = torch.from_numpy(pcm).cuda()
torch.cuda.synchronize()
cap = 8 * n + 65536
for rep in range(3):
    t0 = time.perf_counter(); enc.analyze_device(d.data_ptr(), stride, n); t1 = time.perf_counter()
    a = enc.pack(cap, on_device=True); t2 = time.perf_counter()
    b = enc.pack(cap, on_device=False); t3 = time.perf_counter()
    c = enc.encode_whole(pcm); t4 = time.perf_counter()
    assert a == b == c
    if rep == 0:
        outbuf = np.zeros(cap, np.uint8)
        enc.encode_whole(pcm, out=outbuf)
    t5 = time.perf_counter(); v = enc.encode_whole(pcm, out=outbuf); t6 = time.perf_counter()
    assert bytes(v) == a
    print("   encode_whole into a reused output buffer: %.2f ms = %.0f Msamples/s" % ((t6 - t5) * 1e3, n / (t6 - t5) / 1e6))
    print("n=%d analyze %.2f ms | device pack %.2f ms | host pack %.2f ms | encode_whole (H2D+analyze+device pack) %.2f ms = %.0f Msamples/s"
          % (n, (t1 - t0) * 1e3, (t2 - t1) * 1e3, (t3 - t2) * 1e3, (t4 - t3) * 1e3, n / (t4 - t3) / 1e6))
