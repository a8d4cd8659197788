"""End-to-end SLAEncoder_EncodeWhole from caller memory (pageable and page-locked), with the handle's phase trace.
usage: SLA_HIP_TRACE=1 python tests/tools/e2e_trace.py CFG SECONDS
Measurement tool, not a test."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
import torch          # noqa: E402

import bench          # noqa: E402
import sla_amd        # noqa: E402


def main():
    cfg = sys.argv[1]
    nch, bits, rate, seconds, order, ltm, lms, ms, win, maxb, cap = bench.CONFIGS[cfg]
    if len(sys.argv) > 2 and sys.argv[2]:
        seconds = int(sys.argv[2])
    n = rate * seconds
    host = bench.synth_device(torch, nch, n, bits, rate, 0, n).cpu()
    pcm = host.numpy()
    enc = sla_amd.Encoder(*cap)
    enc.set_wave_format(nch, bits, rate)
    enc.set_encode_parameter(order, ltm, lms, ms, win, maxb)
    out = np.zeros(4 * nch * n + 65536, np.uint8)
    pin_in = host.pin_memory()
    pin_out = torch.zeros(4 * nch * n + 65536, dtype=torch.uint8).pin_memory()
    ref = None
    modes = [("pageable plain", pcm, out, {"stream": 0}), ("pinned plain", pin_in.numpy(), pin_out.numpy(), {"stream": 0}),
             ("pageable default", pcm, out, {"stream": 1}), ("pinned default", pin_in.numpy(), pin_out.numpy(), {"stream": 1})]
    for name, src, dst, opts in modes:
        for k, v in opts.items():
            enc.set_option(k, v)
        for rep in range(4):
            sys.stderr.write("---- %s rep %d\n" % (name, rep))
            t0 = time.perf_counter()
            data = enc.encode_whole(src, out=dst)
            dt = time.perf_counter() - t0
            if ref is None:
                ref = bytes(data)
            print("%s %ds %s rep %d: %.2f ms = %.0f Msamples/s, %d bytes, same=%s"
                  % (cfg, seconds, name, rep, dt * 1e3, n * nch / dt / 1e6, len(data), bytes(data) == ref), flush=True)


if __name__ == "__main__":
    main()
