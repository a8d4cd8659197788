"""A/B of option "upload24" (VERDICT r2 item 8): SLAEncoder_EncodeWhole from PAGEABLE host memory, 24-bit material,
three bytes per sample on the bus (k_unpack24) against the int32 planes as they are.  Interleaved rounds on one box.
    python tests/tools/pack24_ab.py [C3|C5] [seconds] > profiles/r3_pack24_ab_<cfg>.json"""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
import torch          # noqa: E402

import bench          # noqa: E402
import sla_amd        # noqa: E402


def main():
    cfg = sys.argv[1] if len(sys.argv) > 1 else "C3"
    nch, bits, rate, seconds, order, ltm, lms, ms, win, maxb, cap = bench.CONFIGS[cfg]
    if len(sys.argv) > 2:
        seconds = int(sys.argv[2])
    n = rate * seconds
    pcm = np.ascontiguousarray(bench.synth_device(torch, nch, n, bits, rate, 0, n).cpu().numpy())
    torch.cuda.empty_cache()
    out = np.zeros(min(4 * nch * n + 65536, 0xFFFFFFF0), np.uint8)
    res = {"config": cfg, "seconds": seconds, "samples_x_channels": n * nch, "rounds": []}
    encs = {}
    for mode in (0, 1):
        e = sla_amd.Encoder(*cap)
        e.set_wave_format(nch, bits, rate)
        e.set_encode_parameter(order, ltm, lms, ms, win, maxb)
        e.set_option("upload24", mode)
        encs[mode] = e
    ref = None
    for rnd in range(4):
        row = {}
        for mode in (0, 1):
            e = encs[mode]
            for stream in (1, 0):
                e.set_option("stream", stream)
                data = e.encode_whole(pcm, out=out)
                if ref is None:
                    ref = bytes(data)
                assert bytes(data) == ref, "upload24 changed the bytes"
                t0 = time.perf_counter()
                e.encode_whole(pcm, out=out)
                dt = time.perf_counter() - t0
                row["%s_%s" % ("upload24" if mode else "int32", "streamed" if stream else "plain")] = round(n * nch / dt / 1e6, 1)
        res["rounds"].append(row)
    keys = sorted(res["rounds"][0])
    res["median_msamples_s"] = {k: float(np.median([r[k] for r in res["rounds"][1:]])) for k in keys}
    res["note"] = ("Msamples/s of SLAEncoder_EncodeWhole, pageable caller memory -> .sla bytes in caller memory; first round = warm-up, "
                   "median of the other three; the bytes are identical in every cell")
    print(json.dumps(res, indent=1))
    for e in encs.values():
        e.close()


if __name__ == "__main__":
    main()
