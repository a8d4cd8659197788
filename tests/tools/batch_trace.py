"""timeline of sla_hip_analyze_batch_device on BASELINE C4's batch (125 ten-second 48 kHz 16-bit stereo clips resident in HBM;
dev tool): SLA_HIP_TRACE=1 python tests/tools/batch_trace.py [clips]"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import torch
import sla_amd
import slalibs as S
nclips = int(sys.argv[1]) if len(sys.argv) > 1 else 125
clip_n, tile = 480000, 1024
pitch = (clip_n + tile - 1) // tile * tile
distinct = [S.synth_pcm(2, clip_n, 16, 48000, seed=4000 + k) for k in range(8)]
d_pcm = torch.zeros((2, nclips * pitch), dtype=torch.int32, device="cuda")
for k in range(nclips):
    d_pcm[:, k * pitch:k * pitch + clip_n] = torch.from_numpy(distinct[k % 8]).cuda()
starts = np.arange(nclips, dtype=np.uint32) * pitch
lens = np.full(nclips, clip_n, np.uint32)
enc = sla_amd.Encoder(2, 4096, 16, 1, 8)
enc.set_wave_format(2, 16, 48000)
enc.set_encode_parameter(16, 1, 8, 1, 1, 4096)
torch.cuda.synchronize()
quiet = os.dup(2)
for _ in range(3):
    enc.analyze_batch_device(d_pcm.data_ptr(), nclips * pitch, nclips * pitch, starts, lens)
print("---- traced step", file=sys.stderr, flush=True)
t = time.perf_counter()
for _ in range(5):
    tm, _ = enc.analyze_batch_device(d_pcm.data_ptr(), nclips * pitch, nclips * pitch, starts, lens)
dt = (time.perf_counter() - t) / 5
print("per batch %.3f ms = %.1f Msamples/s; timing %s; expand %s" % (dt * 1e3, 2 * nclips * clip_n / dt / 1e6, [round(x, 3) for x in tm], enc.last_expand()))
