"""Step time of the analysis for several pipeline settings, same process, same inputs; the settings take turns
(ROUNDS rounds of STEPS steps each) and the median round is reported, so clock drift hits them alike.
usage: python tests/tools/chunk_sweep.py CFG SECONDS [steps] [rounds]
Measurement tool, not a test: prints one line per setting."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
import torch          # noqa: E402

import bench          # noqa: E402
import sla_amd        # noqa: E402

SETTINGS = [
    {},
    {"tail_taps": 1},
    {"tail_taps": 2},
    {"tail_taps": 4},
    {"tail_taps": 1, "tail_waves": 4},
    {"tail_taps": 2, "tail_waves": 4},
    {"tail_taps": 4, "tail_waves": 4},
]
DEFAULTS = {"chunks": 1, "first_chunk": 0, "single_tail": 1, "device_ltm": 1, "tail_taps": 0, "tail_waves": 0, "alt_streams": 2, "lpc_pack": 0, "lpc_blocks_chains": 0, "lpc_threads": 0, "lpc_tile": 0}


def main():
    global SETTINGS
    if os.environ.get("SWEEP_SETTINGS"):                  # a JSON list of option dicts instead of the built-in list
        import json
        SETTINGS = json.loads(os.environ["SWEEP_SETTINGS"])
    cfg = sys.argv[1]
    nch, bits, rate, seconds, order, ltm, lms, ms, win, maxb, cap = bench.CONFIGS[cfg]
    if len(sys.argv) > 2 and sys.argv[2]:
        seconds = int(sys.argv[2])
    steps = int(sys.argv[3]) if len(sys.argv) > 3 else 10
    rounds = int(sys.argv[4]) if len(sys.argv) > 4 else 7
    n = rate * seconds
    stride = (n + 63) // 64 * 64
    d_pcm = torch.zeros((nch, stride), dtype=torch.int32, device="cuda")
    d_pcm[:, :n] = bench.synth_device(torch, nch, n, bits, rate, 0, n)
    d_lat = torch.zeros_like(d_pcm)
    d_fin = torch.zeros_like(d_pcm)
    enc = sla_amd.Encoder(*cap)
    enc.set_wave_format(nch, bits, rate)
    enc.set_encode_parameter(order, ltm, lms, ms, win, maxb)
    enc.bind_residual_planes(d_lat.data_ptr(), d_fin.data_ptr(), stride)
    torch.cuda.synchronize()
    ref = None
    times = np.zeros((len(SETTINGS), rounds))
    parts = np.zeros((len(SETTINGS), 12))
    same = [True] * len(SETTINGS)
    for r in range(rounds):
        for i, st in enumerate(SETTINGS):
            for k, v in {**DEFAULTS, **st}.items():
                enc.set_option(k, v)
            for _ in range(2):
                enc.analyze_device(d_pcm.data_ptr(), stride, n)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(steps):
                parts[i] += np.array(enc.analyze_device(d_pcm.data_ptr(), stride, n))
            torch.cuda.synchronize()
            times[i, r] = (time.perf_counter() - t0) / steps * 1e3
            chk = int(d_fin.to(torch.int64).sum().item())
            if ref is None:
                ref = chk
            same[i] = same[i] and chk == ref
    parts /= steps * rounds
    for i, st in enumerate(SETTINGS):
        acc = parts[i]
        print("%s %ds %-55s median %.3f min %.3f ms/step | search %.2f blocks %.2f lattice %.2f acf %.2f tail %.2f host %.2f/%.2f same=%s"
              % (cfg, seconds, st, np.median(times[i]), times[i].min(), acc[1], acc[2], acc[3], acc[8], acc[4], acc[5], acc[6], same[i]), flush=True)


if __name__ == "__main__":
    main()
