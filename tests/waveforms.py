"""Deterministic test inputs.

`gen(name, ...)` gives the eight waveform families the reference's round-trip suite uses
(reference test/test_SLAEncodeDecode.c:57-187: silence, 440 Hz sine, white noise, chirp, +/- full
scale constants, Nyquist oscillation, Gaussian noise), re-stated with numpy's seeded generators
(the reference seeds libc rand(), which is not reproducible across libcs), quantised the way the
reference's tests quantise doubles to left-justified int32 (test_SLAEncodeDecode.c:270-290)."""
import numpy as np

NAMES = ("silence", "sine", "white", "chirp", "posconst", "negconst", "nyquist", "gauss")


def _unit(name, nch, n, seed):
    rng = np.random.default_rng(seed)
    t = np.arange(n, dtype=np.float64)
    if name == "silence":
        x = np.zeros((nch, n))
    elif name == "sine":
        x = np.tile(np.sin(440.0 * 2 * np.pi * t / 44100.0), (nch, 1))
    elif name == "white":
        x = 2.0 * (rng.random((nch, n)) - 0.5)
    elif name == "chirp":
        x = np.tile(np.sin((2.0 * np.pi * t) / (n - t)), (nch, 1))
    elif name == "posconst":
        x = np.ones((nch, n))
    elif name == "negconst":
        x = -np.ones((nch, n))
    elif name == "nyquist":
        x = np.tile(np.where(t % 2 == 0, 1.0, -1.0), (nch, 1))
    elif name == "gauss":
        x = np.clip(0.25 * rng.standard_normal((nch, n)), -1.0, 1.0)
    else:
        raise KeyError(name)
    return x


def gen(name, nch, n, bits, lshift=0, seed=0):
    """planar left-justified int32 [nch][n] with `lshift` extra low zero bits inside the bits."""
    x = _unit(name, nch, n, seed)
    eff = bits - lshift
    full = float(1 << (eff - 1))
    q = np.clip(np.rint(x * full), -full, full - 1).astype(np.int64)
    return np.ascontiguousarray(((q << (32 - eff)).astype(np.int64)).astype(np.int32))


def music_like(nch, n, bits, seed=1, level=0.5):
    """AR(2)-coloured noise + two partials: compresses like programme material."""
    rng = np.random.default_rng(seed)
    out = np.zeros((nch, n))
    t = np.arange(n, dtype=np.float64)
    for ch in range(nch):
        e = rng.standard_normal(n) * 0.02
        y = np.zeros(n)
        a1, a2 = 1.6 - 0.05 * ch, -0.8
        for i in range(2, n):
            y[i] = a1 * y[i - 1] + a2 * y[i - 2] + e[i]
        y = y / (np.abs(y).max() + 1e-9)
        out[ch] = level * (0.6 * y + 0.25 * np.sin(2 * np.pi * (220.0 + 3 * ch) * t / 48000.0)
                           + 0.1 * np.sin(2 * np.pi * 3521.0 * t / 48000.0))
    full = float(1 << (bits - 1))
    q = np.clip(np.rint(out * full), -full, full - 1).astype(np.int64)
    return np.ascontiguousarray(((q << (32 - bits)).astype(np.int64)).astype(np.int32))
