"""repeat encodes with changing sizes/parameters and watch device + host memory (dev tool)"""
import os, sys, resource
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..")); sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
import numpy as np, torch
torch.cuda.init()
import sla_amd, waveforms as W
rng = np.random.default_rng(3)
def mem():
    free, total = torch.cuda.mem_get_info()
    return (total - free) / 2**20, resource.getrusage(resource.RUSAGE_SELF).ru_maxrss / 1024
enc = sla_amd.Encoder(2, 8192, 32, 3, 8)
for it in range(300):
    nch = int(rng.integers(1, 3)); bits = int(rng.choice([16, 24])); n = int(rng.integers(1000, 400000))
    pcm = W.music_like(nch, n, bits, seed=it)
    enc.set_wave_format(nch, bits, 48000)
    enc.set_encode_parameter(int(rng.choice([8, 16, 32])), int(rng.choice([1, 3])), 8, 0, int(rng.integers(0, 5)), int(rng.choice([2048, 4096, 8192])))
    enc.encode_whole(pcm)
    if it % 50 == 0 or it == 299:
        print(it, "device MiB used %.0f, host maxrss MiB %.0f" % mem(), flush=True)
enc.close()
for it in range(40):      # create / destroy
    e2 = sla_amd.Encoder(2, 4096, 16, 1, 8); e2.set_wave_format(1, 16, 48000); e2.set_encode_parameter(16, 1, 8, 0, 1, 4096)
    e2.encode_whole(W.music_like(1, 50000, 16, seed=it)); e2.close()
print("after create/destroy x40: device MiB used %.0f, host maxrss MiB %.0f" % mem())
