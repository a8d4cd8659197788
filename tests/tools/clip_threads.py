"""SLAEncoder_EncodeWhole clip by clip from T host threads, one encoder handle each (dev tool): how far do the fixed
per-call latencies of different handles overlap on one GPU?  python clip_threads.py [threads ...]"""
import os, sys, time, threading
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import sla_amd
import slalibs as S
clips = [S.synth_pcm(2, 480000, 16, 48000, seed=4000 + k) for k in range(8)]
def worker(enc, n, out):
    buf = np.zeros(4 * 2 * 480000 + 65536, np.uint8)
    for k in range(n):
        enc.encode_whole(clips[k % 8], out=buf)
for T in [int(a) for a in sys.argv[1:]] or [1, 2, 4, 8]:
    encs = []
    for _ in range(T):
        e = sla_amd.Encoder(2, 4096, 16, 1, 8); e.set_wave_format(2, 16, 48000); e.set_encode_parameter(16, 1, 8, 1, 1, 4096)
        e.set_option("threads", 1)
        encs.append(e)
    for e in encs: worker(e, 2, None)
    n = 40
    th = [threading.Thread(target=worker, args=(e, n, None)) for e in encs]
    t0 = time.perf_counter()
    for t in th: t.start()
    for t in th: t.join()
    dt = time.perf_counter() - t0
    print("threads %d: %.3f ms per clip overall, %.1f Msamples/s" % (T, dt / (n * T) * 1e3, n * T * 960000 / dt / 1e6), flush=True)
    for e in encs: e.close()
