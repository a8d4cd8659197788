"""how many super-frames the device plan hands back to the host, per kind of material (dev tool):
python tests/tools/plan_stats.py  ->  kind, super-frames, host-planned, chunks from device tables"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import torch
torch.cuda.init()
import sla_amd
import slalibs as S
import waveforms as W


def run(name, pcm, nch, bits, rate, order, ms, maxb):
    enc = sla_amd.Encoder(nch, maxb, order, 1, 8)
    enc.set_option("stream", 0)
    enc.set_wave_format(nch, bits, rate)
    enc.set_encode_parameter(order, 1, 8, ms, 1, maxb)
    enc.encode_whole(pcm)
    c = enc.last_counters()
    nsf = (pcm.shape[1] + maxb - 1) // maxb
    print("%-28s frames %6d  host-planned %5d  chain groups %5d  expand %s  block cert %s" % (name, nsf, c[1], c[0], enc.last_expand(), enc.last_block_cert()), flush=True)
    enc.close()


n = 48000 * 120
run("music 16 stereo ms", W.music_like(2, n, 16, seed=1), 2, 16, 48000, 16, 1, 4096)
run("music 24 stereo ms o32", W.music_like(2, n, 24, seed=2), 2, 24, 48000, 32, 1, 4096)
run("music 24 mono o48 8192", W.music_like(1, n, 24, seed=3), 1, 24, 96000, 48, 0, 8192)
run("bench 16 mono", S.synth_pcm(1, n, 16, 48000, seed=4), 1, 16, 48000, 16, 0, 4096)
run("bench 24 stereo o32", S.synth_pcm(2, n, 24, 48000, seed=5), 2, 24, 48000, 32, 1, 4096)
rng = np.random.default_rng(6)
x = (rng.integers(-(1 << 20), 1 << 20, size=(2, n), dtype=np.int64) << 8).astype(np.int32)
run("noise 24 stereo", x, 2, 24, 48000, 16, 1, 4096)
t = np.arange(n)
tone = (np.sin(2 * np.pi * 440.0 / 48000 * t) * (1 << 22) + rng.normal(0, 2.0, n)).astype(np.int64)
run("tone+floor 24 mono o32", (tone[None, :] << 8).astype(np.int32), 1, 24, 48000, 32, 0, 4096)
run("music 16 stereo o16 gaps", S.synth_pcm(2, n, 16, 48000, seed=8, gaps=True), 2, 16, 48000, 16, 1, 4096)
