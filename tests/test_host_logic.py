"""CPU-only tests of the product's host logic (no compute on a GPU is attempted here):
the C-ABI library loads and exports every declared symbol, and the scalar host pieces that sit
between kernel launches (window tables, code-length estimate, shortest path, silence hops,
long-term solve, bit-serial block packer, file header) agree bit-for-bit with the oracle."""
import ctypes as C
import os
import re

import numpy as np
import pytest

import sla_amd
import slalibs as S
import waveforms as W

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
u8p, u32p, i32p, f64p = sla_amd.u8p, sla_amd.u32p, sla_amd.i32p, sla_amd.f64p


@pytest.fixture(scope="module")
def L():
    if not os.path.exists(sla_amd.LIB_PATH):
        sla_amd.build()
    return sla_amd.lib()


def ptr(a, t):
    return a.ctypes.data_as(t)


def test_library_exports_every_declared_symbol(L):
    declared = set()
    for hdr in ("SLAEncoder.h", "sla_hip.h", "SLAPredictor.h", "SLACoder.h", "SLADecoder.h"):
        text = re.sub(r"/\*.*?\*/", " ", open(os.path.join(ROOT, "include", hdr)).read(), flags=re.S)     # prototypes only, not prose
        declared |= set(re.findall(r"\b(SLA[A-Z][A-Za-z]*_\w+|sla_hip_\w+)\s*\(", text))
    assert declared == set(sla_amd.EXPORTED_SYMBOLS)
    for name in sorted(declared):
        assert hasattr(L, name), name


def test_struct_layouts_match_reference_abi():
    """field order / sizes of the public structs (reference src/include/public/SLA.h:61-86)"""
    assert C.sizeof(sla_amd.SLAWaveFormat) == 16
    assert C.sizeof(sla_amd.SLAEncodeParameter) == 24
    assert C.sizeof(sla_amd.SLAEncoderConfig) == 24
    assert C.sizeof(sla_amd.SLAHeaderInfo) == 56
    assert sla_amd.SLAWaveFormat.offset_lshift.offset == 12
    assert sla_amd.SLAEncodeParameter.max_num_block_samples.offset == 20


def test_no_cpu_fallback_without_device(L):
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    with pytest.raises(RuntimeError):
        sla_amd.Encoder()


def test_null_arguments(L):
    assert L.SLAEncoder_SetWaveFormat(None, None) == 2
    assert L.SLAEncoder_SetEncodeParameter(None, None) == 2
    assert L.SLAEncoder_EncodeHeader(None, None, 0) == 2
    assert L.SLAEncoder_EncodeWhole(None, None, 0, None, 0, None) == 2
    assert L.SLAEncoder_EncodeBlock(None, None, 0, None, 0, None) == 2
    L.SLAEncoder_Destroy(None)


def test_header_bytes(L, oracle):
    pcm = W.music_like(2, 5000, 16)
    p = S.make_params(2, 16, 44100, 16, 1, 8, 1, 1, 4096)
    _, data, tr = oracle.encode_trace(p, pcm)
    hdr = sla_amd.encode_header(2, 16, 44100, tr.offset_lshift, 16, 1, 8, 1, 1, 4096, 5000, tr.num_blocks,
                                int(tr.blk_bytes[:tr.num_blocks].max()),
                                int(max(8 * int(b) * 44100 // int(n) for b, n in
                                        zip(tr.blk_bytes[:tr.num_blocks], tr.blk_nsmpl[:tr.num_blocks]))))
    assert hdr == data[:43]
    small = np.zeros(8, np.uint8)
    h = sla_amd.SLAHeaderInfo()
    assert L.SLAEncoder_EncodeHeader(C.byref(h), ptr(small, u8p), 8) == 4


def test_decode_header_host_side(L, oracle):
    """SLADecoder_DecodeHeader is host-only: every field of a header written by the oracle comes back, a damaged
    header is reported but still delivered, wrong magic / version / size are refused (src/SLADecoder.c:171-251)"""
    pcm = W.music_like(2, 5000, 24, seed=3)
    p = S.make_params(2, 24, 96000, 32, 3, 16, 1, 1, 8192)
    _, data, tr = oracle.encode_trace(p, pcm)
    rc, h = sla_amd.decode_header(data)
    assert rc == 0
    assert (h.wave_format.num_channels, h.wave_format.bit_per_sample, h.wave_format.sampling_rate,
            h.wave_format.offset_lshift) == (2, 24, 96000, tr.offset_lshift)
    assert (h.encode_param.parcor_order, h.encode_param.longterm_order, h.encode_param.lms_order_per_filter,
            h.encode_param.ch_process_method, h.encode_param.max_num_block_samples) == (32, 3, 16, 1, 8192)
    assert (h.num_samples, h.num_blocks) == (5000, tr.num_blocks)
    assert h.max_block_size == int(tr.blk_bytes[:tr.num_blocks].max())
    _, _, hdr = oracle.decode_whole(p, data, 5000)
    assert h.max_block_size == int(hdr[11]) and h.max_bit_per_second == int.from_bytes(data[39:43], "big")
    bad = bytearray(data); bad[16] ^= 4
    rc, h2 = sla_amd.decode_header(bytes(bad))
    assert rc == 11 and h2.num_samples != 5000 and h2.num_blocks == tr.num_blocks
    bad = bytearray(data); bad[2] = ord("+")
    assert sla_amd.decode_header(bytes(bad))[0] == 10
    bad = bytearray(data); bad[13] = 2
    assert sla_amd.decode_header(bytes(bad))[0] == 10
    assert sla_amd.decode_header(data[:42])[0] == 9
    assert L.SLADecoder_DecodeHeader(None, 43, C.byref(h)) == 2
    # without a device there is no decoder either
    import torch
    if not torch.cuda.is_available():
        cfg = sla_amd.SLADecoderConfig(2, 4096, 16, 1, 8, 1, 0)
        assert L.SLADecoder_Create(C.byref(cfg)) is None


@pytest.mark.parametrize("wtype", range(5))
@pytest.mark.parametrize("n", [1, 2, 2048, 3000, 4096])
def test_window_tables(L, oracle, wtype, n):
    w = np.zeros(n)
    assert L.slai_make_window(wtype, ptr(w, f64p), n) == 0
    assert np.array_equal(w.view(np.uint64), oracle.window(wtype, n).view(np.uint64))


def test_code_length(L, oracle):
    L.slai_code_length.restype = C.c_double
    L.slai_code_length.argtypes = [C.c_double, C.c_uint32, C.c_uint32, f64p, C.c_uint32]
    for name in W.NAMES:
        for bits, order in ((16, 16), (24, 32)):
            x = W.gen(name, 1, 3072, bits, seed=4)[0].astype(np.float64) * 2.0 ** -31
            _, par = oracle.parcor(x, order)
            r0 = oracle.autocorr(x, 1)[0]
            got = L.slai_code_length(r0, 3072, bits, ptr(par, f64p), order)
            assert got.hex() == oracle.code_length(x, bits, par).hex()


def test_shortest_path(L, oracle):
    BIG = float(1 << 24)
    rng = np.random.default_rng(0)
    for nodes in (2, 5, 9, 17):
        for _ in range(20):
            adj = np.full((nodes, nodes), BIG)
            for i in range(nodes):
                for j in range(i + 1, nodes):
                    if rng.random() < 0.5 or j == i + 1:
                        adj[i, j] = float(rng.integers(1, 6))      # many ties: first-minimum rule matters
            path = np.zeros(nodes, np.uint32)
            assert L.slai_shortest_path(ptr(adj, f64p), nodes, ptr(path, u32p)) == 0
            _, _, want = oracle.dijkstra(adj, 0, nodes - 1)
            node = nodes - 1
            while node != 0:                                       # compare along the chosen route
                assert path[node] == want[node]
                node = int(path[node])
    adj = np.full((3, 3), BIG)
    path = np.zeros(3, np.uint32)
    assert L.slai_shortest_path(ptr(adj, f64p), 3, ptr(path, u32p)) == -1


def test_zero_run_helpers(L):
    L.slai_zero_run.restype = C.c_uint32
    L.slai_zero_run.argtypes = [C.POINTER(C.c_uint64), C.c_uint64, C.c_uint64]
    rng = np.random.default_rng(1)
    n = 5000
    nzb = rng.random(n) < 0.002
    nzb[700:3000] = False
    words = np.zeros(n // 64 + 3, np.uint64)
    for i in np.nonzero(nzb)[0]:
        words[i // 64] |= np.uint64(1) << np.uint64(i % 64)
    for start in (0, 1, 63, 64, 699, 700, 701, 2999, 4990):
        for limit in (1, 10, 64, 2048, 4096):
            limit = min(limit, n - start)
            ahead = np.nonzero(nzb[start:start + limit])[0]
            want = int(ahead[0]) if len(ahead) else limit
            assert L.slai_zero_run(ptr(words, C.POINTER(C.c_uint64)), start, limit) == want


@pytest.mark.parametrize("ntaps", [1, 3, 5])
def test_longterm_solve(L, oracle, ntaps):
    """Toeplitz solve + stability fallback from the compact record the FFT kernel hands back
    (the FFT and the peak scan run on the GPU and are covered by the -m gpu tests)"""
    L.slai_ltm_solve.argtypes = [f64p, C.c_uint32, u32p, f64p]
    rng = np.random.default_rng(ntaps)
    cases = [W.gen(nm, 1, 4096, 16, seed=11)[0] >> 18 for nm in W.NAMES]
    base = rng.integers(-2000, 2000, 97)
    cases.append((np.tile(base, 50)[:4096] + rng.integers(-50, 50, 4096)).astype(np.int32))
    cases.append((np.tile(base[:2], 2048) * 3).astype(np.int32))
    checked = 0
    for fft in (8192, 32768):
        for res in cases:
            ret, pitch, coef, ac = oracle.ltm_analyze(np.ascontiguousarray(res, np.int32), fft, ntaps, want_autocorr=True)
            if ret != 0:
                continue
            rec = np.zeros(12)
            if abs(ac[0]) <= np.finfo(np.float32).tiny:
                rec[0] = 0.0
            else:
                rec[0], rec[1] = 1.0, float(pitch)
                rec[2:7] = ac[:5]
                rec[7:12] = [ac[pitch + k - 2] if pitch + k >= 2 else 0.0 for k in range(5)]
            got_pitch = C.c_uint32(0)
            got = np.zeros(5)
            assert L.slai_ltm_solve(ptr(rec, f64p), ntaps, C.byref(got_pitch), ptr(got, f64p)) == 0
            assert got_pitch.value == pitch
            assert np.array_equal(got[:ntaps].view(np.uint64), coef.view(np.uint64))
            checked += 1
    assert checked >= 4
    rec = np.zeros(12); rec[0] = 2.0
    assert L.slai_ltm_solve(ptr(rec, f64p), ntaps, C.byref(C.c_uint32(0)), ptr(np.zeros(5), f64p)) == 4
    rec[0], rec[1] = 1.0, float(ntaps // 2)            # chosen lag too close to the origin
    assert L.slai_ltm_solve(ptr(rec, f64p), ntaps, C.byref(C.c_uint32(0)), ptr(np.zeros(5), f64p)) == 4


def test_fft_twiddle_tables(L, oracle):
    """the twiddle tables shipped to k_ltm_acf reproduce the reference FFT when driven by a plain
    numpy butterfly loop (same evaluation order as the kernel)"""
    L.slai_fft_plan_create.restype = C.c_void_p
    L.slai_fft_plan_create.argtypes = [C.c_uint32]
    L.slai_fft_plan_destroy.argtypes = [C.c_void_p]
    L.slai_fft_plan_export.argtypes = [C.c_void_p, f64p]
    F = 1024
    plan = L.slai_fft_plan_create(F)
    tw = np.zeros(6 * F)
    L.slai_fft_plan_export(plan, ptr(tw, f64p))
    L.slai_fft_plan_destroy(plan)
    rng = np.random.default_rng(0)
    x = rng.standard_normal(F)

    def stages(z, twr, twi):
        npts = F // 2
        mmax = 2
        while mmax < 2 * npts:
            h = mmax // 2
            for b in range(npts // 2):
                k, blk = b % h, b // h
                i = 2 * k + blk * 2 * mmax
                q = i + mmax
                wr, wi = twr[h - 1 + k], twi[h - 1 + k]
                tr = wr * z[q] - wi * z[q + 1]
                ti = wr * z[q + 1] + wi * z[q]
                z[q], z[q + 1], z[i], z[i + 1] = z[i] - tr, z[i + 1] - ti, z[i] + tr, z[i + 1] + ti
            mmax *= 2

    def real_pass(d, c2, rtr, rti):
        for i in range(2, F // 4 + 1):
            i1 = 2 * i - 2; i2 = i1 + 1; i3 = F - i1; i4 = i3 + 1
            wr, wi = rtr[i - 2], rti[i - 2]
            h1r = 0.5 * (d[i1] + d[i3]); h1i = 0.5 * (d[i2] - d[i4])
            h2r = -c2 * (d[i2] + d[i4]); h2i = c2 * (d[i1] - d[i3])
            d[i1] = h1r + wr * h2r - wi * h2i; d[i2] = h1i + wr * h2i + wi * h2r
            d[i3] = h1r - wr * h2r + wi * h2i; d[i4] = -h1i + wr * h2i + wi * h2r

    npts, lg = F // 2, 9
    rev = [int(format(k, "0%db" % lg)[::-1], 2) for k in range(npts)]
    d = np.zeros(F)
    for k in range(npts):
        d[2 * rev[k]], d[2 * rev[k] + 1] = x[2 * k], x[2 * k + 1]
    stages(d, tw[0:F // 2], tw[F // 2:F])
    real_pass(d, -0.5, tw[2 * F:], tw[2 * F + F // 4:])
    h = d[0]; d[0], d[1] = h + d[1], h - d[1]
    assert np.array_equal(d.view(np.uint64), oracle.fft(x, 1).view(np.uint64))
    real_pass(d, 0.5, tw[2 * F + 2 * (F // 4):], tw[2 * F + 3 * (F // 4):])
    h = d[0]; d[0], d[1] = 0.5 * (h + d[1]), 0.5 * (h - d[1])
    e = d.copy()
    for k in range(npts):
        e[2 * rev[k]], e[2 * rev[k] + 1] = d[2 * k], d[2 * k + 1]
    stages(e, tw[F:F + F // 2], tw[F + F // 2:2 * F])
    assert np.array_equal(e.view(np.uint64), oracle.fft(oracle.fft(x, 1), -1).view(np.uint64))


class BlockParams(C.Structure):
    _fields_ = [(n, C.c_uint32) for n in ("num_samples", "type", "num_channels", "order", "ntaps", "bps",
                                          "lshift", "mid_side")] + \
               [("code", i32p), ("rshift", u32p), ("pitch", u32p), ("ltm_q", i32p), ("rice_init", u32p),
                ("res", i32p * 8)]


PACK_CASES = [
    ("music", 1, 16, 16, 1, 8, 0), ("music", 2, 24, 32, 3, 8, 1), ("synth_gaps", 2, 16, 16, 1, 8, 1),
    ("white", 2, 16, 16, 1, 8, 1), ("sine", 8, 16, 8, 1, 4, 0), ("posconst", 1, 24, 16, 5, 16, 0),
]


@pytest.mark.parametrize("kind,nch,bits,order,ltm,lms,ms", PACK_CASES)
def test_block_packer_reproduces_oracle_bytes(L, oracle, kind, nch, bits, order, ltm, lms, ms):
    """every block of an oracle encode, re-packed by the product's packer from the traced
    parameters + residuals, must give exactly the oracle's bytes (header fields, Rice/Golomb/gamma
    body, RAW body, size and CRC16 patch-up)"""
    n = 12000
    if kind == "music":
        pcm = W.music_like(nch, n, bits, seed=3)
    elif kind == "synth_gaps":
        pcm = S.synth_pcm(nch, n, bits, gaps=True)
        pcm[:, :3000] = 0
    else:
        pcm = W.gen(kind, nch, n, bits, seed=3)
    p = S.make_params(nch, bits, 48000, order, ltm, lms, ms, 1, 4096)
    ret, data, tr = oracle.encode_trace(p, pcm)
    assert ret == 0
    L.slai_pack_block.restype = C.c_uint32
    off = 43
    shift = 32 - bits + tr.offset_lshift
    for b in range(tr.num_blocks):
        s, cnt, typ = int(tr.blk_start[b]), int(tr.blk_nsmpl[b]), int(tr.blk_type[b])
        code = np.ascontiguousarray(tr.code[b])
        ltm_q = np.zeros((nch, 5), np.int32)
        ltm_q[:, :tr.ltm_coef.shape[2]] = tr.ltm_coef[b]
        rshift = np.ascontiguousarray(tr.rshift[b])
        pitch = np.ascontiguousarray(tr.pitch[b])
        rice = np.ascontiguousarray(tr.rice_init[b])
        if typ == 2:
            xi = pcm[:, s:s + cnt] >> shift
            if ms:
                l, r = xi[0].astype(np.int64), xi[1].astype(np.int64)
                xi = np.stack([((l + r) >> 1), (l - r)]).astype(np.int32)
            planes = [np.ascontiguousarray(xi[c]) for c in range(nch)]
        else:
            planes = [np.ascontiguousarray(tr.res_final[c, s:s + cnt]) for c in range(nch)]
        bp = BlockParams(cnt, typ, nch, order, ltm, bits, tr.offset_lshift, ms,
                         ptr(code, i32p), ptr(rshift, u32p), ptr(pitch, u32p), ptr(ltm_q, i32p), ptr(rice, u32p),
                         (i32p * 8)(*([ptr(a, i32p) for a in planes] + [None] * (8 - nch))))
        out = np.zeros(int(tr.blk_bytes[b]) + 64, np.uint8)
        size = L.slai_pack_block(C.byref(bp), ptr(out, u8p), len(out))
        assert size == tr.blk_bytes[b], (b, typ)
        assert out[:size].tobytes() == data[off:off + size], (b, typ)
        if size > 16:
            assert L.slai_pack_block(C.byref(bp), ptr(out, u8p), size - 1) == 0   # too small -> 0
        off += size
    assert off == len(data)


def test_crc16_known_answers(L):
    """reference test_SLAUtility.c:41-46, 71-72"""
    L.slai_crc16.restype = C.c_uint32
    L.slai_crc16.argtypes = [C.c_char_p, C.c_size_t]
    assert L.slai_crc16(b"123456789", 9) == 0xBB3D
    data = open(os.path.join(ROOT, "tests", "golden", "a.wav"), "rb").read()
    assert L.slai_crc16(data, len(data)) == 0xA611
