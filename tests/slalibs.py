"""ctypes bindings shared by the test-suite, bench.py and __graft_entry__.

Three libraries:
  * ``oracle``  -- oracle/libsla_oracle.so, this repo's CPU restatement (test infrastructure)
  * ``ref``     -- oracle/_ref/libsla_ref.so, the UNMODIFIED reference compiled in the build
                   container (may be absent; never read at run time on the GPU box unless the
                   prebuilt .so travelled there)
  * product     -- sla_amd (see sla_amd/__init__.py); not loaded from here.

All three checkers expose the same flat signatures with a different prefix
(``slao_`` for the oracle, ``ref_`` for the reference probe).
"""
import ctypes as C
import os
import subprocess
import wave

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")

u8p = C.POINTER(C.c_uint8)
u32p = C.POINTER(C.c_uint32)
i32p = C.POINTER(C.c_int32)
f64p = C.POINTER(C.c_double)


class FlatParams(C.Structure):
    _fields_ = [(n, C.c_uint32) for n in (
        "num_channels", "bits_per_sample", "sampling_rate",
        "parcor_order", "longterm_order", "lms_order",
        "ch_process_method", "window_type", "max_block_samples",
        "cap_channels", "cap_block_samples", "cap_parcor_order",
        "cap_longterm_order", "cap_lms_order")]


class FlatTrace(C.Structure):
    _fields_ = [
        ("max_blocks", C.c_uint32), ("order_stride", C.c_uint32),
        ("ltm_stride", C.c_uint32), ("sample_stride", C.c_uint32),
        ("num_blocks", C.c_uint32), ("offset_lshift", C.c_uint32),
        ("blk_start", u32p), ("blk_nsmpl", u32p), ("blk_type", u32p), ("blk_bytes", u32p),
        ("parcor", f64p), ("code", i32p), ("kint", i32p),
        ("rshift", u32p), ("pitch", u32p), ("ltm_coef", i32p), ("rice_init", u32p),
        ("res_lattice", i32p), ("res_final", i32p)]


def make_params(num_channels=1, bits=16, rate=48000, parcor=16, ltm=1, lms=8, ms=0, window=1,
                max_block=4096, cap=None):
    """cap = (channels, block_samples, parcor, ltm, lms); default = the CLI's capacity
    (reference src/main.c:94-99)."""
    if cap is None:
        cap = (8, 16384, 48, 5, 40)
    return FlatParams(num_channels, bits, rate, parcor, ltm, lms, ms, window, max_block, *cap)


def _ptr(a, typ):
    return a.ctypes.data_as(typ)


class Trace:
    """numpy-backed sla_flat_trace."""

    def __init__(self, params, num_samples, max_blocks=None):
        Cn = params.num_channels
        O = params.parcor_order + 1
        L = max(params.longterm_order, 1)
        if max_blocks is None:
            max_blocks = num_samples // 1024 + 8
        self.C, self.O, self.L = Cn, O, L
        z = lambda shape, dt: np.zeros(shape, dtype=dt)
        self.blk_start = z(max_blocks, np.uint32)
        self.blk_nsmpl = z(max_blocks, np.uint32)
        self.blk_type = z(max_blocks, np.uint32)
        self.blk_bytes = z(max_blocks, np.uint32)
        self.parcor = z((max_blocks, Cn, O), np.float64)
        self.code = z((max_blocks, Cn, O), np.int32)
        self.kint = z((max_blocks, Cn, O), np.int32)
        self.rshift = z((max_blocks, Cn), np.uint32)
        self.pitch = z((max_blocks, Cn), np.uint32)
        self.ltm_coef = z((max_blocks, Cn, L), np.int32)
        self.rice_init = z((max_blocks, Cn), np.uint32)
        self.res_lattice = z((Cn, max(num_samples, 1)), np.int32)
        self.res_final = z((Cn, max(num_samples, 1)), np.int32)
        self.c = FlatTrace(
            max_blocks, O, L, max(num_samples, 1), 0, 0,
            _ptr(self.blk_start, u32p), _ptr(self.blk_nsmpl, u32p), _ptr(self.blk_type, u32p),
            _ptr(self.blk_bytes, u32p), _ptr(self.parcor, f64p), _ptr(self.code, i32p),
            _ptr(self.kint, i32p), _ptr(self.rshift, u32p), _ptr(self.pitch, u32p),
            _ptr(self.ltm_coef, i32p), _ptr(self.rice_init, u32p),
            _ptr(self.res_lattice, i32p), _ptr(self.res_final, i32p))

    @property
    def num_blocks(self):
        return int(self.c.num_blocks)

    @property
    def offset_lshift(self):
        return int(self.c.offset_lshift)


class CheckerLib:
    """Uniform python face over libsla_oracle.so (prefix slao_) / libsla_ref.so (prefix ref_)."""

    def __init__(self, path, prefix):
        self.lib = C.CDLL(path)
        self.prefix = prefix
        self.path = path

    def fn(self, name, restype=C.c_int):
        f = getattr(self.lib, self.prefix + name)
        f.restype = restype
        return f

    # -- unit level ---------------------------------------------------------
    def autocorr(self, x, nlags):
        x = np.ascontiguousarray(x, np.float64)
        r = np.zeros(nlags, np.float64)
        self.fn("autocorr")(_ptr(x, f64p), C.c_uint32(len(x)), _ptr(r, f64p), C.c_uint32(nlags))
        return r

    def levinson(self, r, order):
        r = np.ascontiguousarray(r, np.float64)
        lpc = np.zeros(order + 2, np.float64)
        par = np.zeros(order + 2, np.float64)
        ret = self.fn("levinson")(_ptr(r, f64p), C.c_uint32(order), _ptr(lpc, f64p), _ptr(par, f64p))
        return ret, lpc[:order + 1], par[:order + 1]

    def parcor(self, x, order):
        x = np.ascontiguousarray(x, np.float64)
        par = np.zeros(order + 1, np.float64)
        ret = self.fn("parcor")(_ptr(x, f64p), C.c_uint32(len(x)), C.c_uint32(order), _ptr(par, f64p))
        return ret, par

    def code_length(self, x, bps, parcor):
        x = np.ascontiguousarray(x, np.float64)
        parcor = np.ascontiguousarray(parcor, np.float64)
        out = C.c_double(0)
        self.fn("code_length")(_ptr(x, f64p), C.c_uint32(len(x)), C.c_uint32(bps), _ptr(parcor, f64p),
                               C.c_uint32(len(parcor) - 1), C.byref(out))
        return out.value

    def lattice_predict(self, x, kint):
        x = np.ascontiguousarray(x, np.int32)
        kint = np.ascontiguousarray(kint, np.int32)
        res = np.zeros(len(x), np.int32)
        self.fn("lattice_predict")(_ptr(x, i32p), C.c_uint32(len(x)), _ptr(kint, i32p),
                                   C.c_uint32(len(kint) - 1), _ptr(res, i32p))
        return res

    def lattice_synth(self, res, kint):
        res = np.ascontiguousarray(res, np.int32)
        kint = np.ascontiguousarray(kint, np.int32)
        out = np.zeros(len(res), np.int32)
        self.fn("lattice_synth")(_ptr(res, i32p), C.c_uint32(len(res)), _ptr(kint, i32p),
                                 C.c_uint32(len(kint) - 1), _ptr(out, i32p))
        return out

    def preemph_i32(self, x):
        x = np.array(x, np.int32)
        self.fn("preemph_i32")(_ptr(x, i32p), C.c_uint32(len(x)))
        return x

    def deemph_i32(self, x):
        x = np.array(x, np.int32)
        self.fn("deemph_i32")(_ptr(x, i32p), C.c_uint32(len(x)))
        return x

    def preemph_f64(self, x):
        x = np.array(x, np.float64)
        self.fn("preemph_f64", None)(_ptr(x, f64p), C.c_uint32(len(x)))
        return x

    def ltm_analyze(self, res, fft_size, ntaps, max_taps=5, want_autocorr=False):
        res = np.ascontiguousarray(res, np.int32)
        pitch = C.c_uint32(0)
        coef = np.zeros(max_taps, np.float64)
        ac = np.zeros(fft_size, np.float64) if want_autocorr else None
        ret = self.fn("ltm_analyze")(_ptr(res, i32p), C.c_uint32(len(res)), C.c_uint32(fft_size),
                                     C.c_uint32(max_taps), C.c_uint32(ntaps), C.byref(pitch),
                                     _ptr(coef, f64p), _ptr(ac, f64p) if want_autocorr else None)
        return (ret, pitch.value, coef[:ntaps], ac) if want_autocorr else (ret, pitch.value, coef[:ntaps])

    def _ltm(self, name, x, pitch, coef):
        x = np.ascontiguousarray(x, np.int32)
        coef = np.ascontiguousarray(coef, np.int32)
        out = np.zeros(len(x), np.int32)
        self.fn(name)(_ptr(x, i32p), C.c_uint32(len(x)), C.c_uint32(pitch), _ptr(coef, i32p),
                      C.c_uint32(len(coef)), _ptr(out, i32p))
        return out

    def ltm_predict(self, x, pitch, coef):
        return self._ltm("ltm_predict", x, pitch, coef)

    def ltm_synth(self, x, pitch, coef):
        return self._ltm("ltm_synth", x, pitch, coef)

    def _lms(self, name, x, order):
        x = np.ascontiguousarray(x, np.int32)
        out = np.zeros(len(x), np.int32)
        self.fn(name)(_ptr(x, i32p), C.c_uint32(len(x)), C.c_uint32(order), _ptr(out, i32p))
        return out

    def lms_predict(self, x, order):
        return self._lms("lms_predict", x, order)

    def lms_synth(self, x, order):
        return self._lms("lms_synth", x, order)

    def partition_search(self, data, min_blk, delta, max_blk, bps, order):
        data = np.ascontiguousarray(data, np.float64)
        nch, n = data.shape
        nparts = C.c_uint32(0)
        parts = np.zeros(n // delta + 4, np.uint32)
        ret = self.fn("partition_search")(_ptr(data, f64p), C.c_uint32(nch), C.c_uint32(n), C.c_uint32(min_blk),
                                          C.c_uint32(delta), C.c_uint32(max_blk), C.c_uint32(bps),
                                          C.c_uint32(order), C.byref(nparts), _ptr(parts, u32p))
        return ret, parts[:nparts.value].copy()

    def dijkstra(self, adjacency, start, goal):
        adjacency = np.ascontiguousarray(adjacency, np.float64)
        nodes = adjacency.shape[0]
        cost = C.c_double(0)
        path = np.zeros(nodes, np.uint32)
        ret = self.fn("dijkstra")(_ptr(adjacency, f64p), C.c_uint32(nodes), C.c_uint32(start), C.c_uint32(goal),
                                  C.byref(cost), _ptr(path, u32p))
        return ret, cost.value, path

    def crc16(self, data):
        data = np.frombuffer(bytes(data), np.uint8)
        return int(self.fn("crc16", C.c_uint32)(_ptr(data, u8p), C.c_uint32(len(data))))

    def fft(self, data, sign):
        data = np.array(data, np.float64)
        self.fn("fft", None)(_ptr(data, f64p), C.c_uint32(len(data)), C.c_int32(sign))
        return data

    def window(self, wtype, n):
        w = np.zeros(n, np.float64)
        self.fn("window")(C.c_uint32(wtype), _ptr(w, f64p), C.c_uint32(n))
        return w

    def bitwidth(self, x):
        x = np.ascontiguousarray(x, np.int32)
        return int(self.fn("bitwidth", C.c_uint32)(_ptr(x, i32p), C.c_uint32(len(x))))

    def lesolve(self, A, b, iters=2):
        A = np.ascontiguousarray(A, np.float64)
        b = np.array(b, np.float64)
        ret = self.fn("lesolve")(_ptr(A, f64p), _ptr(b, f64p), C.c_uint32(len(b)), C.c_uint32(iters))
        return ret, b

    def rice_init(self, res):
        res = np.ascontiguousarray(res, np.int32)
        nch, n = res.shape
        out = np.zeros(nch, np.uint32)
        self.fn("rice_init", None)(_ptr(res, i32p), C.c_uint32(nch), C.c_uint32(n), _ptr(out, u32p))
        return out

    def code_residual(self, res, bps):
        res = np.ascontiguousarray(res, np.int32)
        nch, n = res.shape
        cap = 16 * nch * n + 1024
        out = np.zeros(cap, np.uint8)
        size = self.fn("code_residual", C.c_uint32)(_ptr(res, i32p), C.c_uint32(nch), C.c_uint32(n), C.c_uint32(bps),
                                                    _ptr(out, u8p), C.c_uint32(cap))
        return out[:size].tobytes()

    def decode_residual(self, data, nch, n, bps):
        buf = np.frombuffer(bytes(data) + b"\0" * 16, np.uint8)
        res = np.zeros((nch, n), np.int32)
        self.fn("decode_residual", None)(_ptr(buf, u8p), C.c_uint32(len(buf)), C.c_uint32(nch), C.c_uint32(n),
                                         C.c_uint32(bps), _ptr(res, i32p))
        return res

    # -- codec level --------------------------------------------------------
    def encode_whole(self, params, pcm):
        pcm = np.ascontiguousarray(pcm, np.int32)
        nch, n = pcm.shape
        cap = 8 * nch * n + 65536
        out = np.zeros(cap, np.uint8)
        size = C.c_uint32(0)
        ret = self.fn("encode_whole")(C.byref(params), _ptr(pcm, i32p), C.c_uint32(n), _ptr(out, u8p),
                                      C.c_uint32(cap), C.byref(size))
        return ret, out[:size.value].tobytes()

    def encode_range(self, params, pcm, file_lshift):
        """a range of a longer file whose offset_lshift is `file_lshift` (oracle only; multi-GPU sharding tests)"""
        pcm = np.ascontiguousarray(pcm, np.int32)
        nch, n = pcm.shape
        cap = 8 * nch * n + 65536
        out = np.zeros(cap, np.uint8)
        size = C.c_uint32(0)
        ret = self.fn("encode_range")(C.byref(params), _ptr(pcm, i32p), C.c_uint32(n), C.c_uint32(file_lshift),
                                      _ptr(out, u8p), C.c_uint32(cap), C.byref(size))
        return ret, out[:size.value].tobytes()

    def encode_fixed_blocks(self, params, pcm, block_samples):
        pcm = np.ascontiguousarray(pcm, np.int32)
        nch, n = pcm.shape
        cap = 8 * nch * n + 65536
        out = np.zeros(cap, np.uint8)
        size = C.c_uint32(0)
        ret = self.fn("encode_fixed_blocks")(C.byref(params), _ptr(pcm, i32p), C.c_uint32(n),
                                             C.c_uint32(block_samples), _ptr(out, u8p), C.c_uint32(cap),
                                             C.byref(size))
        return ret, out[:size.value].tobytes()

    def encode_trace(self, params, pcm):
        pcm = np.ascontiguousarray(pcm, np.int32)
        nch, n = pcm.shape
        cap = 8 * nch * n + 65536
        out = np.zeros(cap, np.uint8)
        size = C.c_uint32(0)
        tr = Trace(params, n)
        ret = self.fn("encode_trace")(C.byref(params), _ptr(pcm, i32p), C.c_uint32(n), _ptr(out, u8p),
                                      C.c_uint32(cap), C.byref(size), C.byref(tr.c))
        return ret, out[:size.value].tobytes(), tr

    def decode_whole(self, params, data, nmax):
        buf = np.frombuffer(bytes(data), np.uint8)
        nch = buf[14] if len(buf) > 14 else params.num_channels
        out = np.zeros((max(int(nch), 1), max(nmax, 1)), np.int32)
        ns = C.c_uint32(0)
        hdr = np.zeros(12, np.uint32)
        ret = self.fn("decode_whole")(C.byref(params), _ptr(buf, u8p), C.c_uint32(len(buf)), _ptr(out, i32p),
                                      C.c_uint32(max(nmax, 1)), C.byref(ns), _ptr(hdr, u32p))
        return ret, out[:, :ns.value].copy(), hdr


def build_oracle():
    """Compile oracle/libsla_oracle.so (and oracle/_ref when /root/reference is present)."""
    subprocess.run(["make", "-C", ORACLE_DIR, "-s", "all"], check=True)
    subprocess.run(["make", "-C", ORACLE_DIR, "-s", "ref"], check=True)


_cache = {}


def oracle():
    if "oracle" not in _cache:
        path = os.path.join(ORACLE_DIR, "libsla_oracle.so")
        if not os.path.exists(path):
            build_oracle()
        _cache["oracle"] = CheckerLib(path, "slao_")
    return _cache["oracle"]


def ref():
    """The compiled reference, or None when it was never built (e.g. fresh clone on a GPU box)."""
    if "ref" not in _cache:
        path = os.path.join(ORACLE_DIR, "_ref", "libsla_ref.so")
        _cache["ref"] = CheckerLib(path, "ref_") if os.path.exists(path) else None
    return _cache["ref"]


def parcor_same(tg, to, nb, comp=None, gslice=None):
    """PARCOR doubles of a HIP trace against the oracle's trace: bit patterns for the (block, channel) pairs the exact
    chain kernels analysed (tg.parcor_exact == 1); for the certified ones a quarter of a quantisation step -- 2^-17 for
    the three 16-bit coefficients, 2^-9 for the 8-bit ones -- which is the widest radius the certificate admits (their
    codes, kint and RAW decisions are compared bit for bit by the callers; tests/test_gpu_cert.py checks how far inside
    the certificate's own bound the doubles sit).  gslice: the HIP trace's block range (default [0, nb))."""
    gs = gslice if gslice is not None else slice(0, nb)
    a, b = tg.parcor[gs], to.parcor[:nb]
    ex = tg.parcor_exact[gs].astype(bool)
    if comp is not None:
        a, b, ex = a[comp], b[comp], ex[comp]
    if a.size == 0:
        return True
    tol = np.full(a.shape[-1], 2.0 ** -9)
    tol[:4] = 2.0 ** -17
    bits_ok = np.array_equal(a.view(np.uint64)[ex], b.view(np.uint64)[ex])
    near_ok = bool(np.all(np.abs(a[~ex] - b[~ex]) <= tol))
    return bits_ok and near_ok


# ---- deterministic inputs -------------------------------------------------

def read_wav(path):
    """RIFF PCM -> planar left-justified int32, the layout of reference src/wav.c:392-416."""
    with wave.open(path, "rb") as w:
        nch, width, rate, n = w.getnchannels(), w.getsampwidth(), w.getframerate(), w.getnframes()
        raw = w.readframes(n)
    if width == 1:
        a = (np.frombuffer(raw, np.uint8).astype(np.int32) - 128) << 24
    elif width == 2:
        a = np.frombuffer(raw, "<i2").astype(np.int32) << 16
    elif width == 3:
        b = np.frombuffer(raw, np.uint8).reshape(-1, 3).astype(np.int32)
        a = ((b[:, 0] << 8) | (b[:, 1] << 16) | (b[:, 2] << 24)).astype(np.int32)
    else:
        a = np.frombuffer(raw, "<i4").astype(np.int32)
    return np.ascontiguousarray(a.reshape(-1, nch).T), width * 8, rate


def synth_pcm(num_channels, num_samples, bits, rate=48000, seed=12345, gaps=False):
    """BASELINE.md's synthetic generator: three sines + uniform LCG noise, rounded to `bits`,
    left-justified in int32.  `gaps` inserts 0.5 s of digital silence every 2 s."""
    t = np.arange(num_samples, dtype=np.float64) / rate
    out = np.zeros((num_channels, num_samples), np.int32)
    state = np.uint64(seed)
    full = float(1 << (bits - 1))
    for ch in range(num_channels):
        # LCG s = s*1664525 + 1013904223 (mod 2^32), vectorised by jumping
        idx = np.arange(1, num_samples + 1, dtype=np.uint64)
        s = np.empty(num_samples, np.uint32)
        cur = int(state) & 0xFFFFFFFF
        a, c = 1664525, 1013904223
        # plain loop in chunks keeps it exact and fast enough via numpy cumulative trick
        # (affine maps compose: precompute powers)
        A = np.empty(num_samples, np.uint64)
        Cc = np.empty(num_samples, np.uint64)
        aa, cc = a, c
        A[0], Cc[0] = aa, cc
        # doubling construction
        filled = 1
        while filled < num_samples:
            m = min(filled, num_samples - filled)
            Af, Cf = int(A[filled - 1]), int(Cc[filled - 1])
            A[filled:filled + m] = (A[:m] * np.uint64(Af)) & np.uint64(0xFFFFFFFF)
            Cc[filled:filled + m] = (A[:m] * np.uint64(Cf) + Cc[:m]) & np.uint64(0xFFFFFFFF)
            filled += m
        s = ((A * np.uint64(cur) + Cc) & np.uint64(0xFFFFFFFF)).astype(np.uint32)
        state = np.uint64(int(s[-1]))
        noise = ((s >> 8).astype(np.float64) / float(1 << 24) - 0.5) * 2.0 * 0.02
        x = (0.35 * np.sin(2 * np.pi * 220.0 * (ch + 1) * t)
             + 0.2 * np.sin(2 * np.pi * 1333.7 * t + ch)
             + 0.1 * np.sin(2 * np.pi * 5011.3 * t) + noise)
        q = np.clip(np.rint(x * full), -full, full - 1).astype(np.int64)
        if gaps:
            period, gap = 2 * rate, rate // 2
            mask = (np.arange(num_samples) % period) < gap
            q[mask] = 0
        out[ch] = (q << (32 - bits)).astype(np.int64).astype(np.int32)
    return out
