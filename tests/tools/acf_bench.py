"""Micro-benchmark of the long-term autocorrelation launcher (sla_hip_launch_ltm_acf) on synthetic residuals:
k_ltm_acf (one LDS pass per step) against k_ltm_acf2.
    python tests/tools/acf_bench.py [block_len] [jobs]"""
import ctypes as C
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
import sla_amd  # noqa: E402


class Tuning(C.Structure):
    _fields_ = [("lpc_pack", C.c_uint32), ("lpc_threads", C.c_uint32), ("lpc_blocks_chains", C.c_uint32), ("tail_waves", C.c_uint32),
                ("lpc_tile", C.c_uint32), ("tail_lanes", C.c_uint32), ("plan_margin", C.c_double), ("acf_classic", C.c_uint32),
                ("pad_", C.c_uint32)]


class Job(C.Structure):
    _fields_ = [("blk_off", C.c_uint64), ("blk_len", C.c_uint32), ("channel", C.c_uint32)]


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
    jobs = int(sys.argv[2]) if len(sys.argv) > 2 else 11250
    F = 2 * n
    L = sla_amd.lib()
    L.slai_fft_plan_create.restype = C.c_void_p
    L.slai_fft_plan_create.argtypes = [C.c_uint32]
    L.slai_fft_plan_export.argtypes = [C.c_void_p, C.POINTER(C.c_double)]
    plan = L.slai_fft_plan_create(F)
    tw = np.zeros(6 * F)
    L.slai_fft_plan_export(plan, tw.ctypes.data_as(C.POINTER(C.c_double)))
    d_tw = torch.from_numpy(tw).cuda()
    stride = n * jobs
    res = torch.randint(-2000, 2000, (1, stride), dtype=torch.int32, device="cuda")
    ja = (Job * jobs)(*[Job(k * n, n, 0) for k in range(jobs)])
    d_jobs = torch.frombuffer(bytearray(bytes(ja)), dtype=torch.uint8).cuda()
    out = torch.zeros(jobs * 12, dtype=torch.float64, device="cuda")
    L.sla_hip_launch_ltm_acf.argtypes = [C.c_void_p, C.c_uint64, C.c_void_p, C.c_uint32, C.c_uint32, C.c_void_p, C.c_void_p, C.c_uint32,
                                         C.c_void_p, C.c_uint32, C.c_void_p]
    L.sla_hip_use_tuning.argtypes = [C.POINTER(Tuning)]
    stream = torch.cuda.current_stream().cuda_stream

    def run(classic, ablate, reps=5):
        t = Tuning()
        t.acf_classic = classic
        L.sla_hip_use_tuning(C.byref(t))
        for _ in range(2):
            rc = L.sla_hip_launch_ltm_acf(res.data_ptr(), stride, d_jobs.data_ptr(), jobs, F, d_tw.data_ptr(), None, 0, out.data_ptr(), 12, stream)
            assert rc == 0, rc
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(reps):
            L.sla_hip_launch_ltm_acf(res.data_ptr(), stride, d_jobs.data_ptr(), jobs, F, d_tw.data_ptr(), None, 0, out.data_ptr(), 12, stream)
        b.record()
        torch.cuda.synchronize()
        return a.elapsed_time(b) / reps

    print("block %d, fft %d, %d jobs" % (n, F, jobs))
    print("classic                       %.3f ms" % run(1, 0))
    print("k_ltm_acf2                    %.3f ms" % run(0, 0))


if __name__ == "__main__":
    main()
