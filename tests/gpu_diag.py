#!/usr/bin/env python3
"""Stage-by-stage comparison of one HIP encode against the oracle (prints the first mismatch of
every stage).  Debug helper for the GPU box:  python tests/gpu_diag.py [case]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import sla_amd            # noqa: E402
import slalibs as S       # noqa: E402
import waveforms as W     # noqa: E402


def diag(name, pcm, bits, rate, order, ltm, lms, ms, win, maxb, cap):
    o = S.oracle()
    p = S.make_params(pcm.shape[0], bits, rate, order, ltm, lms, ms, win, maxb, cap=cap)
    ret, want, to = o.encode_trace(p, pcm)
    enc = sla_amd.Encoder(*cap)
    enc.set_wave_format(pcm.shape[0], bits, rate)
    enc.set_encode_parameter(order, ltm, lms, ms, win, maxb)
    t0 = time.time()
    got = enc.encode_whole(pcm)
    dt = time.time() - t0
    tg = enc.trace()
    ok = (got == want)
    print("== %s: bytes %s (%d vs %d) blocks %d vs %d lshift %d vs %d  %.1f ms" % (
        name, "IDENTICAL" if ok else "DIFFER", len(got), len(want), tg.num_blocks, to.num_blocks,
        tg.offset_lshift, to.offset_lshift, dt * 1e3))
    nb = min(tg.num_blocks, to.num_blocks)
    for f in ("blk_start", "blk_nsmpl", "blk_type", "blk_bytes"):
        a, b = getattr(tg, f)[:nb], getattr(to, f)[:nb]
        if not np.array_equal(a, b):
            i = int(np.nonzero(a != b)[0][0])
            print("   %s first mismatch at block %d: %s vs %s" % (f, i, a[max(0, i - 1):i + 3], b[max(0, i - 1):i + 3]))
    comp = (to.blk_type[:nb] == 0) & (tg.blk_type[:nb] == 0) & (tg.blk_start[:nb] == to.blk_start[:nb]) \
        & (tg.blk_nsmpl[:nb] == to.blk_nsmpl[:nb])
    pa, pb = tg.parcor[:nb].view(np.uint64), to.parcor[:nb].view(np.uint64)
    bad = np.nonzero((pa != pb).any(axis=(1, 2)) & comp)[0]
    print("   parcor bit-mismatch blocks: %d of %d" % (len(bad), int(comp.sum())))
    if len(bad):
        i = int(bad[0])
        d = np.abs(tg.parcor[i] - to.parcor[i]).max()
        print("     block %d max |d| = %g\n     gpu %s\n     ora %s" % (i, d, tg.parcor[i, 0, :5], to.parcor[i, 0, :5]))
    for f in ("code", "kint", "rshift", "pitch", "rice_init", "ltm_coef"):
        a, b = getattr(tg, f)[:nb], getattr(to, f)[:nb]
        m = (a != b).reshape(nb, -1).any(axis=1) & comp
        if f == "ltm_coef":
            m &= (to.pitch[:nb] >= 3).any(axis=1)
        if m.any():
            i = int(np.nonzero(m)[0][0])
            print("   %s mismatch in %d blocks, first %d: gpu %s ora %s" % (f, int(m.sum()), i, a[i].ravel()[:8], b[i].ravel()[:8]))
    for f in ("res_lattice", "res_final"):
        nbad = 0
        first = None
        for b in np.nonzero(comp)[0]:
            s, n = int(to.blk_start[b]), int(to.blk_nsmpl[b])
            a, c = getattr(tg, f)[:, s:s + n], getattr(to, f)[:, s:s + n]
            if not np.array_equal(a, c):
                nbad += 1
                if first is None:
                    ch, idx = [int(v[0]) for v in np.nonzero(a != c)]
                    first = (int(b), ch, idx, a[ch, idx:idx + 4], c[ch, idx:idx + 4])
        print("   %s mismatching blocks: %d %s" % (f, nbad, first if first else ""))
    enc.close()
    return ok


def main():
    which = sys.argv[1] if len(sys.argv) > 1 else "all"
    print("device:", sla_amd.device_name() or "(not yet bound)")
    cases = {
        "c2": lambda: diag("c2 mono16 o16", S.synth_pcm(1, 50000, 16), 16, 48000, 16, 1, 8, 0, 1, 4096, (1, 4096, 16, 1, 8)),
        "c3": lambda: diag("c3 stereo24 o32 ms", S.synth_pcm(2, 30000, 24), 24, 48000, 32, 3, 8, 1, 1, 4096, (2, 4096, 32, 3, 8)),
        "c5": lambda: diag("c5 8ch24 o48", S.synth_pcm(8, 20000, 24, 96000), 24, 96000, 48, 3, 8, 0, 1, 8192, (8, 8192, 48, 3, 8)),
        "music": lambda: diag("music stereo16", W.music_like(2, 30000, 16, seed=5), 16, 48000, 16, 1, 8, 1, 1, 4096, (2, 4096, 16, 1, 8)),
        "gaps": lambda: diag("gaps", _gaps(), 16, 48000, 16, 1, 8, 0, 1, 4096, (1, 4096, 16, 1, 8)),
        "white": lambda: diag("white raw", W.gen("white", 2, 9000, 16, seed=2), 16, 48000, 16, 1, 8, 1, 1, 4096, (2, 4096, 16, 1, 8)),
        "p4": lambda: diag("music o32 16384", W.music_like(1, 40000, 16, seed=8), 16, 48000, 32, 3, 8, 0, 1, 16384, (8, 16384, 48, 5, 40)),
    }
    allok = True
    if which.startswith("matrix:"):
        _, name, nch, bits, lshift = which.split(":")
        nch, bits, lshift = int(nch), int(bits), int(lshift)
        pcm = W.gen(name, nch, 8192 + 517, bits, lshift=lshift, seed=nch * 100 + bits)
        ok = diag(which, pcm, bits, 44100, 4, 1, 4, 0, 1, 16384, (8, 16384, 48, 5, 40))
        return 0 if ok else 1
    for k, fn in cases.items():
        if which in ("all", k):
            allok &= bool(fn())
    print("ALL IDENTICAL" if allok else "SOME DIFFER")
    return 0 if allok else 1


def _gaps():
    pcm = S.synth_pcm(1, 30000, 16, gaps=True)
    pcm[:, :3000] = 0
    pcm[:, 9000:14000] = 0
    return pcm


if __name__ == "__main__":
    sys.exit(main())
