"""micro-benchmark of sla_hip_launch_search_exact on a C2-shaped batch (dev tool, not a test)"""
import ctypes as C
import os
import sys
import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
torch.cuda.init()
import sla_amd  # noqa: E402

L = sla_amd.lib()
order = int(os.environ.get("ORDER", "16"))
window = int(os.environ.get("WINDOW", "4096"))
ngroups = int(os.environ.get("GROUPS", "3516"))
lags = L.sla_hip_search_exact_lags(order)
n = ngroups * window
rng = np.random.default_rng(1)
pcm = (rng.integers(-20000, 20000, n, dtype=np.int64).astype(np.int32) << 16)
nodes = (window + 1023) // 1024 + 1
cand = [(i * 1024, min((j - i) * 1024, window - i * 1024)) for i in range(nodes) for j in range(i + 1, nodes)
        if 2048 <= min((j - i) * 1024, window - i * 1024) <= window]


class Group(C.Structure):
    _fields_ = [("pcm_off", C.c_uint64)] + [(k, C.c_uint32) for k in (
        "num_samples", "channel", "win_off", "int_shift", "cand_first", "cand_count", "slot_first", "pad_")]


groups = (Group * ngroups)(*[Group(g * window, window, 0, 0xFFFFFFFF, 16, 0, len(cand), g * len(cand), 0) for g in range(ngroups)])
d_pcm = torch.from_numpy(pcm).cuda()
d_g = torch.frombuffer(bytearray(bytes(groups)), dtype=torch.uint8).cuda()
d_c = torch.from_numpy(np.array(cand, np.uint32)).cuda()
d_ts = torch.zeros(ngroups * 16 * 2 * lags, dtype=torch.float64, device="cuda")
d_out = torch.zeros(ngroups * len(cand) * (order + 2), dtype=torch.float64, device="cuda")
torch.cuda.synchronize()


def run():
    rc = L.sla_hip_launch_search_exact(C.c_void_p(d_pcm.data_ptr()), C.c_uint64(n), 0, order, C.c_void_p(d_g.data_ptr()), ngroups, window, len(cand),
                                       C.c_void_p(d_c.data_ptr()), C.c_void_p(d_ts.data_ptr()), C.c_void_p(d_out.data_ptr()),
                                       C.c_double(2.0 ** 21), C.c_double(0.0), None)
    assert rc == 0


for _ in range(3):
    run()
torch.cuda.synchronize()
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
a.record()
for _ in range(20):
    run()
b.record()
torch.cuda.synchronize()
print("groups %d window %d cands %d order %d: %.1f us per launch pair" % (ngroups, window, len(cand), order, a.elapsed_time(b) / 20 * 1e3))
