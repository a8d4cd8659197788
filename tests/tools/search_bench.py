"""Micro-benchmark of sla_hip_launch_search_exact on loud 24-bit windows (every window over the exactness limit):
with and without the scratch word that lets the exact-window launch return at once.
    python tests/tools/search_bench.py [groups]"""
import ctypes as C
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
import sla_amd  # noqa: E402


class Group(C.Structure):
    _fields_ = [("pcm_off", C.c_uint64)] + [(k, C.c_uint32) for k in (
        "num_samples", "channel", "win_off", "int_shift", "cand_first", "cand_count", "slot_first", "pad_")]


def main():
    ngroups = int(sys.argv[1]) if len(sys.argv) > 1 else 11250
    order, bits, W_ = 48, 24, 8192
    L = sla_amd.lib()
    L.sla_hip_search_exact_lags.restype = C.c_uint32
    lags = L.sla_hip_search_exact_lags(order)
    cand = sorted({(i * 1024, (j - i) * 1024) for i in range(9) for j in range(i + 2, 9)})
    n = W_ * ngroups
    pcm = (torch.randint(-(1 << 22), 1 << 22, (1, n), dtype=torch.int32, device="cuda") << 8)
    groups = (Group * ngroups)(*[Group(g * W_, W_, 0, 0xFFFFFFFF, 32 - bits, 0, len(cand), g * len(cand), 0) for g in range(ngroups)])
    d_g = torch.frombuffer(bytearray(bytes(groups)), dtype=torch.uint8).cuda()
    d_c = torch.from_numpy(np.array(cand, np.uint32)).cuda()
    d_ts = torch.zeros(ngroups * 16 * 2 * lags, dtype=torch.float64, device="cuda")
    d_out = torch.zeros(ngroups * len(cand) * (order + 2), dtype=torch.float64, device="cuda")
    flag = torch.zeros(4, dtype=torch.int32, device="cuda")
    stream = torch.cuda.current_stream().cuda_stream

    def run(use_flag, reps=5):
        args = (C.c_void_p(pcm.data_ptr()), C.c_uint64(n), 0, order, C.c_void_p(d_g.data_ptr()), ngroups, W_, len(cand),
                C.c_void_p(d_c.data_ptr()), C.c_void_p(d_ts.data_ptr()), C.c_void_p(d_out.data_ptr()), C.c_double(2.0 ** 53 * 2.0 ** -62),
                C.c_double(64.0), C.c_void_p(flag.data_ptr()) if use_flag else None, C.c_void_p(stream))
        for _ in range(2):
            assert L.sla_hip_launch_search_exact(*args) == 0
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(reps):
            L.sla_hip_launch_search_exact(*args)
        b.record()
        torch.cuda.synchronize()
        return a.elapsed_time(b) / reps

    print("%d windows of %d samples, order %d" % (ngroups, W_, order))
    print("without the scratch word  %.3f ms" % run(False))
    print("with the scratch word     %.3f ms   (flag now %d)" % (run(True), int(flag[0].item())))


if __name__ == "__main__":
    main()
