import sys, os
sys.path.insert(0, "tests"); sys.path.insert(0, ".")
import numpy as np, torch
torch.cuda.init()
import sla_amd, slalibs as S, waveforms as W
oracle = S.oracle()
for rep in range(3):
    enc = sla_amd.Encoder(2, 8192, 32, 3, 8)
    for i, (nch, bits, order, win, mb, n) in enumerate([(1, 16, 16, 1, 4096, 30000), (2, 24, 32, 2, 8192, 50000), (1, 16, 16, 1, 4096, 9000), (2, 16, 8, 4, 2048, 20000)]):
        pcm = W.music_like(nch, n, bits, seed=i)
        enc.set_wave_format(nch, bits, 48000)
        enc.set_encode_parameter(order, 1, 8, 0, win, mb)
        got = enc.encode_whole(pcm)
        p = S.make_params(nch, bits, 48000, order, 1, 8, 0, win, mb, cap=(2, 8192, 32, 3, 8))
        want = oracle.encode_whole(p, pcm)[1]
        print(rep, i, got == want, len(got), len(want), enc.last_counters(), flush=True)
    enc.close()
