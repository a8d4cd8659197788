"""BASELINE C4 shape: a batch of short stereo clips (48 kHz 16-bit, 10 s, order 16, MS), one SLAEncoder_EncodeWhole
per clip, T host threads with one encoder handle each.  usage: python tests/tools/bench_clips.py [clips] [threads...]"""
import os, sys, time, threading
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import numpy as np
import torch
torch.cuda.init()
import sla_amd
import slalibs as S

clips = int(sys.argv[1]) if len(sys.argv) > 1 else 64
thread_counts = [int(a) for a in sys.argv[2:]] or [1, 2, 4, 8]
n = 480000
base = [S.synth_pcm(2, n, 16, 48000, seed=100 + i) for i in range(8)]
pcms = [base[i % 8] for i in range(clips)]
ref_bytes = {}


def worker(tid, T, out):
    enc = sla_amd.Encoder(2, 4096, 16, 1, 8)
    enc.set_wave_format(2, 16, 48000)
    enc.set_encode_parameter(16, 1, 8, sla_amd.CH_STEREO_MS, sla_amd.WINDOW_SIN, 4096)
    buf = np.zeros(8 * 2 * n + 65536, np.uint8)
    for i in range(tid, clips, T):
        data = enc.encode_whole(pcms[i], out=buf)
        out[i] = hash(data.tobytes())
    enc.close()


enc = sla_amd.Encoder(2, 4096, 16, 1, 8)
enc.set_wave_format(2, 16, 48000)
enc.set_encode_parameter(16, 1, 8, sla_amd.CH_STEREO_MS, sla_amd.WINDOW_SIN, 4096)
single = [enc.encode_whole(pcms[i]) for i in range(8)]
outs = [np.zeros(4 * 2 * n + 65536, np.uint8) for _ in range(clips)]
for rep in range(3):
    t0 = time.perf_counter()
    got = enc.encode_batch(pcms, outs=outs)
    dt = time.perf_counter() - t0
ok = all(rc == 0 and data.tobytes() == single[i % 8] for i, (rc, data) in enumerate(got))
print("sla_hip_encode_batch: %d clips in %.1f ms -> %.3f ms per clip, %.0f Msamples/s (every clip identical to its own EncodeWhole: %s)"
      % (clips, dt * 1e3, dt * 1e3 / clips, clips * n * 2 / dt / 1e6, ok))
enc.close()

for T in thread_counts:
    for rep in range(2):
        out = [None] * clips
        th = [threading.Thread(target=worker, args=(t, T, out)) for t in range(T)]
        t0 = time.perf_counter()
        for t in th:
            t.start()
        for t in th:
            t.join()
        dt = time.perf_counter() - t0
    same = all(out[i] == out[i % 8] for i in range(clips))
    print("threads %d: %d clips in %.1f ms -> %.2f ms per clip, %.0f Msamples/s (identical clips give identical bytes: %s)"
          % (T, clips, dt * 1e3, dt * 1e3 / clips, clips * n * 2 / dt / 1e6, same))
