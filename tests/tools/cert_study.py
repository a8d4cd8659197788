"""Feasibility study for certifying the PARCOR *codes* of the chosen blocks under any-order autocorrelation
sums.  CPU only: the oracle gives the reference-order sums
(reference src/SLAPredictor.c:331-388) and the reference Levinson-Durbin (:253-328); math.fsum gives the
correctly rounded sums an any-order device kernel approximates.

For every analysed block it reports
  * the actual |k_any - k_ref| per coefficient,
  * the first-order bound  eps_m = ||a^(m-1)||_1^2 (1+|k_m|)/e_(m-1) * delta   (see DESIGN 2b),
  * whether any Round(k 2^(q-1)) could flip inside +-SAFETY*eps_m (the block would take the exact path).

Run:  python tests/tools/cert_study.py [--blocks N] [--safety S]
"""
import argparse
import math
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import slalibs  # noqa: E402
import waveforms  # noqa: E402
from certlib import codes, eps_bound, margin  # noqa: E402



def stage(ora, pcm_block, window, bits):
    v = pcm_block.astype(np.float64) * 2.0 ** -31
    return ora.preemph_f64(v * window)


def signals(n_total, rng):
    out = []
    out.append(("bench16", slalibs.synth_pcm(1, n_total, 16)[0], 16))
    out.append(("bench24", slalibs.synth_pcm(1, n_total, 24)[0], 24))
    out.append(("bench24_96k", slalibs.synth_pcm(1, n_total, 24, rate=96000)[0], 24))
    out.append(("music16", waveforms.music_like(1, n_total, 16, seed=3)[0], 16))
    out.append(("music24", waveforms.music_like(1, n_total, 24, seed=4)[0], 24))
    out.append(("white16", waveforms.gen("white", 1, n_total, 16, seed=5)[0], 16))
    out.append(("gauss24", waveforms.gen("gauss", 1, n_total, 24, seed=6)[0], 24))
    out.append(("chirp24", waveforms.gen("chirp", 1, n_total, 24)[0], 24))
    t = np.arange(n_total, dtype=np.float64)
    for db in (-20, -60, -100, -140):
        for bits in (24, 32):
            x = 0.5 * np.sin(2 * np.pi * 997.0 * t / 48000.0) + 0.3 * np.sin(2 * np.pi * 61.0 * t / 48000.0)
            x += (10.0 ** (db / 20.0)) * rng.standard_normal(n_total)
            full = float(1 << (bits - 1))
            q = np.clip(np.rint(x * full), -full, full - 1).astype(np.int64)
            out.append(("tones%+ddB_%d" % (db, bits), ((q << (32 - bits)).astype(np.int64)).astype(np.int32), bits))
    wav = os.path.join(os.path.dirname(__file__), "..", "golden", "a.wav")
    if os.path.exists(wav):
        pcm, bits, _ = slalibs.read_wav(wav)
        out.append(("a.wav", pcm[0], bits))
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--blocks", type=int, default=12)
    ap.add_argument("--safety", type=float, default=16.0)
    args = ap.parse_args()
    ora = slalibs.oracle()
    rng = np.random.default_rng(7)
    print("%-16s %5s %5s | %9s %9s %9s | %8s %6s %6s" % ("signal", "order", "n", "max|dk|", "max ratio", "max eps", "flip", "uncert", "blocks"))
    tot_blocks = tot_uncert = tot_flip = 0
    worst_ratio = 0.0
    for order in (16, 32, 48):
        for n in (4096, 8192):
            win = ora.window(1, n)        # SIN window (reference SLAUtility.c:99-189)
            need = n * args.blocks
            for name, pcm, bits in signals(need + 0, rng):
                nb = min(args.blocks, len(pcm) // n)
                max_dk = max_ratio = max_eps = 0.0
                flips = uncert = 0
                for b in range(nb):
                    x = stage(ora, pcm[b * n:(b + 1) * n], win, bits)
                    r_ref = ora.autocorr(x, order + 1)
                    r_any = np.array([math.fsum(x[:n - lag] * x[lag:]) if lag else math.fsum(x * x) for lag in range(order + 1)])
                    _, _, k_ref = ora.levinson(r_ref, order)
                    _, _, k_any = ora.levinson(r_any, order)
                    if r_any[0] < 1.1920929e-07:
                        continue
                    _, eps1 = eps_bound(r_any, n, order, safety=1.0)
                    bad = False
                    for m in range(1, order + 1):
                        eps = eps1[m]
                        if not np.isfinite(eps):
                            bad = True
                            break
                        dk = abs(k_any[m] - k_ref[m])
                        max_dk = max(max_dk, dk)
                        max_eps = max(max_eps, eps)
                        if eps > 0:
                            max_ratio = max(max_ratio, dk / eps)
                        if margin(k_any[m], m) <= args.safety * eps:
                            bad = True
                    if not np.array_equal(codes(k_any, order), codes(k_ref, order)):
                        flips += 1
                        if not bad:
                            print("  !! UNCAUGHT FLIP", name, order, n, b)
                    uncert += bad
                tot_blocks += nb
                tot_uncert += uncert
                tot_flip += flips
                worst_ratio = max(worst_ratio, max_ratio)
                print("%-16s %5d %5d | %9.2e %9.2e %9.2e | %8d %6d %6d" % (name, order, n, max_dk, max_ratio, max_eps, flips, uncert, nb))
    print("blocks %d, uncertified %d (%.2f %%), real flips %d, worst actual/bound %.3g" % (
        tot_blocks, tot_uncert, 100.0 * tot_uncert / max(tot_blocks, 1), tot_flip, worst_ratio))


if __name__ == "__main__":
    main()
