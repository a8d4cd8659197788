#!/usr/bin/env python3
"""Generates the golden vectors of tests/golden/*.npz from the UNMODIFIED reference compiled into
oracle/_ref (build container only).  Run:  python tests/golden/make_golden.py

A fixture is data only: the input PCM (or the recipe + a hash for the a.wav cases, whose samples
come from the reference's own test fixture) and the reference's outputs for it -- block table,
PARCOR doubles (bit patterns), transmitted codes, lattice coefficients, rshift, pitch, long-term
taps, Rice initial parameters, SHA-1 of both residual planes, and the .sla bytes' size + MD5
(whole bytes for the small cases)."""
import hashlib
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
import slalibs as S          # noqa: E402
import waveforms as W        # noqa: E402

A_WAV = "/root/reference/test/a.wav"

CASES = {
    # name: (input recipe, params kwargs)
    "awav_p0": ("awav", dict(parcor=8, ltm=1, lms=4, ms=0, window=0, max_block=4096)),
    "awav_p2": ("awav", dict(parcor=16, ltm=1, lms=8, ms=0, window=1, max_block=12288)),
    "awav_p4": ("awav", dict(parcor=32, ltm=3, lms=8, ms=0, window=1, max_block=16384)),
    "c2_synth": (("synth", 1, 30000, 16, 48000, False),
                 dict(parcor=16, ltm=1, lms=8, ms=0, window=1, max_block=4096, cap=(1, 4096, 16, 1, 8))),
    "c2_gaps": (("synth", 1, 30000, 16, 48000, True),
                dict(parcor=16, ltm=1, lms=8, ms=0, window=1, max_block=4096, cap=(1, 4096, 16, 1, 8))),
    "c3_synth": (("synth", 2, 20000, 24, 48000, False),
                 dict(parcor=32, ltm=3, lms=8, ms=1, window=1, max_block=4096, cap=(2, 4096, 32, 3, 8))),
    "c4_music": (("music", 2, 20000, 16, 48000, False),
                 dict(parcor=16, ltm=1, lms=8, ms=1, window=1, max_block=4096, cap=(2, 4096, 16, 1, 8))),
    "c5_synth": (("synth", 8, 20000, 24, 96000, False),
                 dict(parcor=48, ltm=3, lms=8, ms=0, window=1, max_block=8192, cap=(8, 8192, 48, 3, 8))),
    "raw_white": (("white", 2, 9000, 16, 48000, False),
                  dict(parcor=16, ltm=1, lms=8, ms=1, window=1, max_block=4096, cap=(2, 4096, 16, 1, 8))),
}


def make_input(recipe):
    if recipe == "awav":
        return S.read_wav(A_WAV)
    kind, nch, n, bits, rate, gaps = recipe
    if kind == "synth":
        pcm = S.synth_pcm(nch, n, bits, rate, gaps=gaps)
        if gaps:
            pcm[:, :3000] = 0
            pcm[:, 9000:14000] = 0
    elif kind == "music":
        pcm = W.music_like(nch, n, bits, seed=5)
    else:
        pcm = W.gen(kind, nch, n, bits, seed=2)
    return pcm, bits, rate


def sha(a):
    return hashlib.sha1(np.ascontiguousarray(a).tobytes()).hexdigest()


def main():
    ref = S.ref()
    assert ref is not None, "build oracle/_ref first (make -C oracle ref)"
    for name, (recipe, kw) in CASES.items():
        pcm, bits, rate = make_input(recipe)
        p = S.make_params(pcm.shape[0], bits, rate, **kw)
        ret, data, tr = ref.encode_trace(p, pcm)
        assert ret == 0
        rd, dec, _ = ref.decode_whole(p, data, pcm.shape[1])
        assert rd == 0 and np.array_equal(dec, pcm)
        nb = tr.num_blocks
        out = dict(
            params=np.array([getattr(p, f) for f, _ in p._fields_], np.uint32),
            input_sha1=sha(pcm), sla_size=len(data), sla_md5=hashlib.md5(data).hexdigest(),
            offset_lshift=tr.offset_lshift,
            blk_start=tr.blk_start[:nb], blk_nsmpl=tr.blk_nsmpl[:nb], blk_type=tr.blk_type[:nb],
            blk_bytes=tr.blk_bytes[:nb], parcor_bits=tr.parcor[:nb].view(np.uint64),
            code=tr.code[:nb], kint=tr.kint[:nb], rshift=tr.rshift[:nb], pitch=tr.pitch[:nb],
            ltm_coef=tr.ltm_coef[:nb], rice_init=tr.rice_init[:nb],
            res_lattice_sha1=sha(tr.res_lattice), res_final_sha1=sha(tr.res_final),
            sla_head=np.frombuffer(data[:4096], np.uint8))
        if recipe != "awav":
            out["pcm"] = pcm
        if len(data) < 70000:
            out["sla"] = np.frombuffer(data, np.uint8)
        np.savez_compressed(os.path.join(HERE, name + ".npz"), **out)
        print(name, pcm.shape, nb, "blocks", len(data), "bytes", out["sla_md5"])

    # unit-level vectors: autocorrelation / PARCOR / lattice on one windowed block
    x = W.music_like(1, 4096, 24, seed=9)[0]
    xd = ref.preemph_f64(x.astype(np.float64) * 2.0 ** -31 * ref.window(1, 4096))
    r = ref.autocorr(xd, 33)
    _, par = ref.parcor(xd, 32)
    xi = x >> 8
    kint = np.zeros(33, np.int32)
    kint[1:] = np.rint(par[1:] * 32767).astype(np.int32) >> 8
    res = ref.lattice_predict(ref.preemph_i32(xi), kint)
    lms = ref.lms_predict(res, 8)
    np.savez_compressed(os.path.join(HERE, "unit_block.npz"), pcm=x, autocorr_bits=r.view(np.uint64),
                        parcor_bits=par.view(np.uint64), kint=kint, lattice=res, lms=lms,
                        rice_init=ref.rice_init(lms[None, :]),
                        code_len_bits=np.array([ref.code_length(xd, 24, par)]).view(np.uint64))
    print("unit_block ok")


if __name__ == "__main__":
    main()
