"""Re-checks the CPU oracle against the committed golden vectors (tests/golden/*.npz, generated
from the unmodified reference by tests/golden/make_golden.py).  Runs everywhere, including the
GPU box where /root/reference does not exist."""
import glob
import hashlib
import os

import numpy as np
import pytest

import slalibs as S

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
CASES = sorted(os.path.basename(f)[:-4] for f in glob.glob(os.path.join(GOLD, "*.npz"))
               if not os.path.basename(f).startswith("unit_"))


def load_case(name):
    g = np.load(os.path.join(GOLD, name + ".npz"))
    p = S.FlatParams(*[int(v) for v in g["params"]])
    if "pcm" in g.files:
        pcm = g["pcm"]
    else:
        pcm, _, _ = S.read_wav(os.path.join(GOLD, "a.wav"))
    assert hashlib.sha1(np.ascontiguousarray(pcm).tobytes()).hexdigest() == str(g["input_sha1"])
    return g, p, pcm


def check_trace_against_golden(g, tr, data):
    """shared with the GPU parity tests: `tr` is any object with the Trace attributes"""
    nb = len(g["blk_start"])
    assert tr.num_blocks == nb and tr.offset_lshift == int(g["offset_lshift"])
    for f in ("blk_start", "blk_nsmpl", "blk_type"):
        assert np.array_equal(getattr(tr, f)[:nb], g[f]), f
    comp = g["blk_type"] == 0
    ex = getattr(tr, "parcor_exact", None)      # HIP trace: 1 where the exact chain kernel ran, 0 where the block was certified
    ex = np.ones_like(comp[:, None] & (tr.rshift[:nb] >= 0)) if ex is None else ex[:nb].astype(bool)
    want = g["parcor_bits"].view(np.float64)
    assert np.array_equal(tr.parcor[:nb].view(np.uint64)[comp & ex.all(axis=1)], g["parcor_bits"][comp & ex.all(axis=1)])
    assert np.all(np.abs(tr.parcor[:nb][comp] - want[comp]) <= 2.0 ** -9)      # certified blocks: the codes below decide
    for f in ("code", "kint", "rshift", "pitch", "rice_init"):
        assert np.array_equal(getattr(tr, f)[:nb][comp], g[f][comp]), f
    used = (g["pitch"] >= 3) & comp[:, None]
    assert np.array_equal(tr.ltm_coef[:nb][used], g["ltm_coef"][used])
    if data is not None:
        assert np.array_equal(tr.blk_bytes[:nb], g["blk_bytes"])
        assert len(data) == int(g["sla_size"])
        assert hashlib.md5(data).hexdigest() == str(g["sla_md5"])


@pytest.mark.parametrize("name", CASES)
def test_oracle_matches_golden(oracle, name):
    g, p, pcm = load_case(name)
    ret, data, tr = oracle.encode_trace(p, pcm)
    assert ret == 0
    check_trace_against_golden(g, tr, data)
    sha = lambda a: hashlib.sha1(np.ascontiguousarray(a).tobytes()).hexdigest()
    assert sha(tr.res_lattice) == str(g["res_lattice_sha1"])
    assert sha(tr.res_final) == str(g["res_final_sha1"])
    rd, dec, _ = oracle.decode_whole(p, data, pcm.shape[1])
    assert rd == 0 and np.array_equal(dec, pcm)


def test_oracle_unit_block(oracle):
    g = np.load(os.path.join(GOLD, "unit_block.npz"))
    x = g["pcm"]
    xd = oracle.preemph_f64(x.astype(np.float64) * 2.0 ** -31 * oracle.window(1, 4096))
    assert np.array_equal(oracle.autocorr(xd, 33).view(np.uint64), g["autocorr_bits"])
    _, par = oracle.parcor(xd, 32)
    assert np.array_equal(par.view(np.uint64), g["parcor_bits"])
    assert np.array([oracle.code_length(xd, 24, par)]).view(np.uint64)[0] == g["code_len_bits"][0]
    res = oracle.lattice_predict(oracle.preemph_i32(x >> 8), g["kint"])
    assert np.array_equal(res, g["lattice"])
    lms = oracle.lms_predict(res, 8)
    assert np.array_equal(lms, g["lms"])
    assert np.array_equal(oracle.rice_init(lms[None, :]), g["rice_init"])


def test_known_answers_from_survey(oracle):
    """a.wav first block at order 8 / RECT / no MS"""
    g, p, pcm = load_case("awav_p0")
    ret, data, tr = oracle.encode_trace(p, pcm)
    assert hashlib.md5(data).hexdigest() == "48c60a59f94f70303be8207d7ea9dc03" and len(data) == 55982
    assert tr.num_blocks == 59 and tr.blk_bytes[0] == 829 and tr.blk_bytes[:59].max() == 1090
    assert list(tr.code[0, 0]) == [0, 27344, 19372, 9283, 28, 13, 9, 9, 4]
    assert list(tr.kint[0, 0]) == [0, 27344, 19372, 9283, 7168, 3328, 2304, 2304, 1024]
    assert tr.parcor[0, 0, 1] == 0.83445743016337226 and tr.parcor[0, 0, 8] == 0.027950627763214334
    assert tr.rice_init[0, 0] == 1 and tr.pitch[0, 0] == 0 and tr.rshift[0, 0] == 0
    assert data[43:67].hex() == "ffff00000337a44b10000356825d612218e06848482002ff"
