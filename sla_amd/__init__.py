"""sla_amd -- MI355X-native encode hot path of the SLA lossless audio codec.

The product is the C-ABI shared library ``sla_amd/libsla_hip.so`` (hand-written gfx950 kernels +
plain-C host, see include/*.h).  This package is only the thin ctypes face used by the tests,
bench.py and __graft_entry__: it mirrors the reference's encoder interface
(SLAEncoder_Create / SetWaveFormat / SetEncodeParameter / EncodeWhole / EncodeBlock, reference
src/include/public/SLAEncoder.h:28-53) and decoder interface (SLADecoder_Create / DecodeHeader /
DecodeWhole, src/include/public/SLADecoder.h:38-60) one to one.

There is no CPU fallback: loading fails loudly when the library has not been built, and
``Encoder()`` raises when no HIP device is usable.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("SLA_HIP_LIB") or os.path.join(_HERE, "libsla_hip.so")     # SLA_HIP_LIB: A/B runs against another build

u8p = C.POINTER(C.c_uint8)
u32p = C.POINTER(C.c_uint32)
i32p = C.POINTER(C.c_int32)
f64p = C.POINTER(C.c_double)

(WINDOW_RECT, WINDOW_SIN, WINDOW_HANN, WINDOW_BLACKMAN, WINDOW_VORBIS) = range(5)
(CH_NONE, CH_STEREO_MS) = range(2)
BLOCK_COMPRESS, BLOCK_SILENT, BLOCK_RAW = 0, 1, 2

API_RESULT = ["OK", "NG", "INVALID_ARGUMENT", "EXCEED_HANDLE_CAPACITY", "INSUFFICIENT_BUFFER_SIZE",
              "INVAILD_CHPROCESSMETHOD", "FAILED_TO_CALCULATE_COEF", "FAILED_TO_PREDICT",
              "FAILED_TO_SYNTHESIZE", "INSUFFICIENT_DATA_SIZE", "INVALID_HEADER_FORMAT",
              "DETECT_DATA_CORRUPTION", "FAILED_TO_FIND_SYNC_CODE", "INVALID_WINDOWFUNCTION_TYPE",
              "NO_DATA_FRAGMENTS", "PARAMETER_NOT_SET"]


class SLAEncoderConfig(C.Structure):
    _fields_ = [("max_num_channels", C.c_uint32), ("max_num_block_samples", C.c_uint32),
                ("max_parcor_order", C.c_uint32), ("max_longterm_order", C.c_uint32),
                ("max_lms_order_per_filter", C.c_uint32), ("verpose_flag", C.c_uint8)]


class SLADecoderConfig(C.Structure):
    _fields_ = [("max_num_channels", C.c_uint32), ("max_num_block_samples", C.c_uint32),
                ("max_parcor_order", C.c_uint32), ("max_longterm_order", C.c_uint32),
                ("max_lms_order_per_filter", C.c_uint32), ("enable_crc_check", C.c_uint8),
                ("verpose_flag", C.c_uint8)]


class SLAStreamingDecoderConfig(C.Structure):
    _fields_ = [("core_config", SLADecoderConfig), ("decode_interval_hz", C.c_float), ("max_bit_per_sample", C.c_uint32)]


class SLAWaveFormat(C.Structure):
    _fields_ = [("num_channels", C.c_uint32), ("bit_per_sample", C.c_uint32),
                ("sampling_rate", C.c_uint32), ("offset_lshift", C.c_uint8)]


class SLAEncodeParameter(C.Structure):
    _fields_ = [("parcor_order", C.c_uint32), ("longterm_order", C.c_uint32),
                ("lms_order_per_filter", C.c_uint32), ("ch_process_method", C.c_int),
                ("window_function_type", C.c_int), ("max_num_block_samples", C.c_uint32)]


class SLAHeaderInfo(C.Structure):
    _fields_ = [("wave_format", SLAWaveFormat), ("encode_param", SLAEncodeParameter),
                ("num_samples", C.c_uint32), ("num_blocks", C.c_uint32),
                ("max_block_size", C.c_uint32), ("max_bit_per_second", C.c_uint32)]


class HipTrace(C.Structure):
    _fields_ = [
        ("max_blocks", C.c_uint32), ("order_stride", C.c_uint32),
        ("ltm_stride", C.c_uint32), ("sample_stride", C.c_uint32),
        ("num_blocks", C.c_uint32), ("offset_lshift", C.c_uint32),
        ("blk_start", u32p), ("blk_nsmpl", u32p), ("blk_type", u32p), ("blk_bytes", u32p),
        ("parcor", f64p), ("code", i32p), ("kint", i32p),
        ("rshift", u32p), ("pitch", u32p), ("ltm_coef", i32p), ("rice_init", u32p),
        ("res_lattice", i32p), ("res_final", i32p), ("parcor_exact", u32p)]


class BatchItem(C.Structure):
    _fields_ = [("input", C.POINTER(i32p)), ("num_samples", C.c_uint32), ("data_size", C.c_uint32),
                ("data", u8p), ("output_size", C.c_uint32), ("result", C.c_int32)]


class SlaError(RuntimeError):
    def __init__(self, code, where):
        name = API_RESULT[code] if 0 <= code < len(API_RESULT) else ("hipError %d" % (-code))
        super().__init__("%s failed: %s (%d)" % (where, name, code))
        self.code = code


def build(verbose=False):
    """Compile sla_amd/libsla_hip.so in-tree (hipcc --offload-arch=gfx950 + gcc)."""
    cmd = ["make", "-C", os.path.join(_HERE, "csrc"), "all"]
    subprocess.run(cmd, check=True, stdout=None if verbose else subprocess.DEVNULL)


_lib = None


def lib():
    """The loaded C-ABI library.  Raises if it has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError("%s is missing: run `python -c 'import __graft_entry__ as g; g.build()'` "
                              "(the HIP path has no CPU fallback)" % LIB_PATH)
        L = C.CDLL(LIB_PATH)
        L.SLAEncoder_Create.restype = C.c_void_p
        L.SLAEncoder_Create.argtypes = [C.POINTER(SLAEncoderConfig)]
        L.SLAEncoder_Destroy.argtypes = [C.c_void_p]
        L.SLAEncoder_Destroy.restype = None
        L.SLAEncoder_SetWaveFormat.argtypes = [C.c_void_p, C.POINTER(SLAWaveFormat)]
        L.SLAEncoder_SetEncodeParameter.argtypes = [C.c_void_p, C.POINTER(SLAEncodeParameter)]
        L.SLAEncoder_EncodeHeader.argtypes = [C.POINTER(SLAHeaderInfo), u8p, C.c_uint32]
        L.SLAEncoder_EncodeWhole.argtypes = [C.c_void_p, C.POINTER(i32p), C.c_uint32, u8p, C.c_uint32, u32p]
        L.SLAEncoder_EncodeBlock.argtypes = [C.c_void_p, C.POINTER(i32p), C.c_uint32, u8p, C.c_uint32, u32p]
        L.sla_hip_analyze_device.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64, C.c_uint32, C.c_void_p,
                                             C.POINTER(C.c_float)]
        L.sla_hip_pack.argtypes = [C.c_void_p, u8p, C.c_uint32, u32p]
        L.sla_hip_pack_device.argtypes = [C.c_void_p, u8p, C.c_uint32, u32p]
        L.sla_hip_get_trace.argtypes = [C.c_void_p, C.POINTER(HipTrace)]
        L.sla_hip_final_residual.restype = C.c_void_p
        L.sla_hip_final_residual.argtypes = [C.c_void_p, C.POINTER(C.c_uint64)]
        L.sla_hip_lattice_residual.restype = C.c_void_p
        L.sla_hip_lattice_residual.argtypes = [C.c_void_p, C.POINTER(C.c_uint64)]
        L.sla_hip_bind_residual_planes.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64]
        L.sla_hip_device_name.restype = C.c_char_p
        L.sla_hip_last_timing.argtypes = [C.c_void_p, C.POINTER(C.c_float)]
        L.sla_hip_last_counters.argtypes = [C.c_void_p, C.POINTER(C.c_uint32)]
        L.sla_hip_last_kernel_ms.argtypes = [C.c_void_p, C.POINTER(C.c_float)]
        L.sla_hip_last_block_cert.argtypes = [C.c_void_p, C.POINTER(C.c_uint32)]
        L.sla_hip_last_cert_audit.argtypes = [C.c_void_p, C.POINTER(C.c_uint32)]
        L.sla_hip_last_expand.argtypes = [C.c_void_p, C.POINTER(C.c_uint32)]
        L.sla_hip_search_exact_lags.restype = C.c_uint32
        L.sla_hip_encoder_set_option.argtypes = [C.c_void_p, C.c_char_p, C.c_double]
        L.sla_hip_shard_scan.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64, C.c_uint32, u32p, C.POINTER(C.c_uint64)]
        L.sla_hip_shard_bounds.argtypes = [C.c_uint32, C.c_uint32, C.POINTER(C.c_uint64), C.c_uint32, u32p]
        L.sla_hip_shard_scan_counts.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64, C.c_uint32, u32p, u32p]
        L.sla_hip_shard_analyze.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64, C.c_uint32, C.c_uint32, C.POINTER(C.c_float)]
        L.sla_hip_shard_analyze_no_silence.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64, C.c_uint32, C.c_uint32, C.POINTER(C.c_float)]
        L.sla_hip_shard_header.argtypes = [C.POINTER(u8p), C.c_uint32, u8p, C.c_uint32]
        L.sla_hip_encode_batch.argtypes = [C.c_void_p, C.POINTER(BatchItem), C.c_uint32]
        L.sla_hip_analyze_batch_device.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64, C.c_uint32, u32p, u32p, C.c_uint32,
                                                   u32p, C.POINTER(C.c_float)]
        L.SLADecoder_Create.restype = C.c_void_p
        L.SLADecoder_Create.argtypes = [C.POINTER(SLADecoderConfig)]
        L.SLADecoder_Destroy.argtypes = [C.c_void_p]
        L.SLADecoder_Destroy.restype = None
        L.SLADecoder_DecodeHeader.argtypes = [u8p, C.c_uint32, C.POINTER(SLAHeaderInfo)]
        L.SLADecoder_SetWaveFormat.argtypes = [C.c_void_p, C.POINTER(SLAWaveFormat)]
        L.SLADecoder_SetEncodeParameter.argtypes = [C.c_void_p, C.POINTER(SLAEncodeParameter)]
        L.SLADecoder_DecodeWhole.argtypes = [C.c_void_p, u8p, C.c_uint32, C.POINTER(i32p), C.c_uint32, u32p]
        L.sla_hip_decoder_last_timing.argtypes = [C.c_void_p, C.POINTER(C.c_float)]
        L.SLAStreamingDecoder_Create.restype = C.c_void_p
        L.SLAStreamingDecoder_Create.argtypes = [C.POINTER(SLAStreamingDecoderConfig)]
        L.SLAStreamingDecoder_Destroy.argtypes = [C.c_void_p]
        L.SLAStreamingDecoder_Destroy.restype = None
        L.SLAStreamingDecoder_SetWaveFormat.argtypes = [C.c_void_p, C.POINTER(SLAWaveFormat)]
        L.SLAStreamingDecoder_SetEncodeParameter.argtypes = [C.c_void_p, C.POINTER(SLAEncodeParameter)]
        for name in ("EstimateMinimumNessesaryDataSize", "EstimateDecodableNumSamples", "GetOutputNumSamplesPerDecode",
                     "GetRemainDataSize"):
            getattr(L, "SLAStreamingDecoder_" + name).argtypes = [C.c_void_p, u32p]
        L.SLAStreamingDecoder_AppendDataFragment.argtypes = [C.c_void_p, u8p, C.c_uint32]
        L.SLAStreamingDecoder_CollectDataFragment.argtypes = [C.c_void_p, C.POINTER(u8p), u32p]
        L.SLAStreamingDecoder_Decode.argtypes = [C.c_void_p, C.POINTER(i32p), C.c_uint32, u32p]
        L.sla_hip_decode_device.argtypes = [C.c_void_p, u8p, C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint64, u32p]
        _lib = L
    return _lib


EXPORTED_SYMBOLS = [
    # include/SLAEncoder.h
    "SLAEncoder_Create", "SLAEncoder_Destroy", "SLAEncoder_SetWaveFormat", "SLAEncoder_SetEncodeParameter",
    "SLAEncoder_EncodeHeader", "SLAEncoder_EncodeBlock", "SLAEncoder_EncodeWhole",
    # include/sla_hip.h
    "sla_hip_launch_prepass", "sla_hip_launch_lpc", "sla_hip_launch_lattice", "sla_hip_launch_tail",
    "sla_hip_launch_ltm_acf", "sla_hip_launch_rice_len", "sla_hip_launch_rice_write", "sla_hip_pack_device", "sla_hip_launch_unpack16", "sla_hip_launch_unpack24",
    "sla_hip_analyze_device", "sla_hip_pack", "sla_hip_final_residual", "sla_hip_lattice_residual",
    "sla_hip_get_trace", "sla_hip_device_name", "sla_hip_last_timing", "sla_hip_launch_search_exact",
    "sla_hip_search_exact_lags", "sla_hip_launch_plan", "sla_hip_last_counters", "sla_hip_launch_lpc_rerun", "sla_hip_last_kernel_ms", "sla_hip_launch_lpc_blocks", "sla_hip_launch_lpc_blocks_cert", "sla_hip_last_block_cert", "sla_hip_last_cert_audit", "sla_hip_launch_batch_scan", "sla_hip_launch_lattice_groups_x", "sla_hip_launch_lpc_blocks_cert_x", "sla_hip_launch_lpc_blocks_x",
    "sla_hip_launch_lpc_x", "sla_hip_launch_ltm_acf_x", "sla_hip_launch_search_exact_x", "sla_hip_launch_tail_x", "sla_hip_last_expand", "sla_hip_launch_expand", "sla_hip_launch_expand_masked",
    "sla_hip_launch_lpc_f64", "sla_hip_launch_lattice_raw", "sla_hip_launch_tail_stages", "sla_hip_launch_emphasis_i32",
    "sla_hip_launch_emphasis_f64", "sla_hip_use_tuning", "sla_hip_launch_lattice_groups", "sla_hip_launch_ltm_solve",
    "sla_hip_encoder_set_option", "sla_hip_shard_scan", "sla_hip_shard_scan_counts", "sla_hip_shard_bounds", "sla_hip_shard_analyze", "sla_hip_shard_analyze_no_silence", "sla_hip_shard_header",
    # include/SLAPredictor.h, include/SLACoder.h (per-call API of the reference, encode side)
    "SLALPCCalculator_Create", "SLALPCCalculator_Destroy", "SLALPCCalculator_CalculatePARCORCoefDouble",
    "SLALPCCalculator_EstimateCodeLength", "SLALPCSynthesizer_Create", "SLALPCSynthesizer_Destroy", "SLALPCSynthesizer_Reset",
    "SLALPCSynthesizer_PredictByParcorCoefInt32", "SLALongTermCalculator_Create", "SLALongTermCalculator_Destroy",
    "SLALongTermCalculator_CalculateCoef", "SLALongTermSynthesizer_Create", "SLALongTermSynthesizer_Destroy",
    "SLALongTermSynthesizer_Reset", "SLALongTermSynthesizer_PredictInt32", "SLALMSFilter_Create", "SLALMSFilter_Destroy",
    "SLALMSFilter_Reset", "SLALMSFilter_PredictInt32", "SLAOptimalEncodeEstimator_Create", "SLAOptimalEncodeEstimator_Destroy",
    "SLAOptimalEncodeEstimator_SearchOptimalBlockPartitions", "SLAOptimalEncodeEstimator_CalculateMaxNumPartitions",
    "SLAEmphasisFilter_Create", "SLAEmphasisFilter_Reset", "SLAEmphasisFilter_Destroy", "SLAEmphasisFilter_PreEmphasisInt32",
    "SLAEmphasisFilter_PreEmphasisDouble", "SLACoder_Create", "SLACoder_Destroy",
    "SLACoder_CalculateInitialRecursiveRiceParameter", "sla_hip_coder_initial_parameter", "sla_hip_bind_residual_planes",
    # include/SLADecoder.h + the decode-side launchers of include/sla_hip.h
    "SLADecoder_DecodeHeader", "SLADecoder_Create", "SLADecoder_Destroy", "SLADecoder_SetWaveFormat",
    "SLADecoder_SetEncodeParameter", "SLADecoder_DecodeWhole", "sla_hip_decoder_last_timing", "sla_hip_decode_device",
    "sla_hip_launch_dec_bits", "sla_hip_launch_dec_lms", "sla_hip_launch_dec_ltm", "sla_hip_launch_dec_lattice",
    "sla_hip_launch_dec_finish", "sla_hip_launch_prepass_tiles", "sla_hip_encode_batch", "sla_hip_analyze_batch_device",
    "SLAStreamingDecoder_Create", "SLAStreamingDecoder_Destroy", "SLAStreamingDecoder_SetWaveFormat",
    "SLAStreamingDecoder_SetEncodeParameter", "SLAStreamingDecoder_EstimateMinimumNessesaryDataSize",
    "SLAStreamingDecoder_EstimateDecodableNumSamples", "SLAStreamingDecoder_GetOutputNumSamplesPerDecode",
    "SLAStreamingDecoder_AppendDataFragment", "SLAStreamingDecoder_CollectDataFragment",
    "SLAStreamingDecoder_GetRemainDataSize", "SLAStreamingDecoder_Decode",
    # decode side of include/SLAPredictor.h
    "SLALPCSynthesizer_SynthesizeByParcorCoefInt32", "SLALongTermSynthesizer_SynthesizeInt32", "SLALMSFilter_SynthesizeInt32",
    "SLAEmphasisFilter_DeEmphasisInt32", "sla_hip_launch_dec_deemphasis",
]


class Trace:
    """numpy view of the per-block results of the last analysis (same fields as the oracle's trace)."""

    def __init__(self, num_channels, order, ltm_order, num_samples, max_blocks, want_residuals=True):
        Cn, O, L = num_channels, order + 1, max(ltm_order, 1)
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
        # 1 = parcor[] are the reference's doubles bit for bit, 0 = certified route (codes / decisions are the reference's)
        self.parcor_exact = z((max_blocks, Cn), np.uint32)
        ns = max(num_samples, 1)
        self.res_lattice = z((Cn, ns), np.int32) if want_residuals else None
        self.res_final = z((Cn, ns), np.int32) if want_residuals else None
        p = lambda a, t: a.ctypes.data_as(t)
        self.c = HipTrace(
            max_blocks, O, L, ns, 0, 0,
            p(self.blk_start, u32p), p(self.blk_nsmpl, u32p), p(self.blk_type, u32p), p(self.blk_bytes, u32p),
            p(self.parcor, f64p), p(self.code, i32p), p(self.kint, i32p), p(self.rshift, u32p),
            p(self.pitch, u32p), p(self.ltm_coef, i32p), p(self.rice_init, u32p),
            p(self.res_lattice, i32p) if want_residuals else None,
            p(self.res_final, i32p) if want_residuals else None,
            p(self.parcor_exact, u32p))

    @property
    def num_blocks(self):
        return int(self.c.num_blocks)

    @property
    def offset_lshift(self):
        return int(self.c.offset_lshift)


class Encoder:
    """Python face of ``struct SLAEncoder`` (reference src/include/public/SLAEncoder.h)."""

    def __init__(self, max_num_channels=8, max_num_block_samples=16384, max_parcor_order=48,
                 max_longterm_order=5, max_lms_order_per_filter=40):
        self._lib = lib()
        cfg = SLAEncoderConfig(max_num_channels, max_num_block_samples, max_parcor_order,
                               max_longterm_order, max_lms_order_per_filter, 0)
        self._h = self._lib.SLAEncoder_Create(C.byref(cfg))
        if not self._h:
            raise RuntimeError("SLAEncoder_Create failed: no usable HIP device (libsla_hip has no CPU fallback)")
        self.num_channels = 0
        self.order = 0
        self.ltm_order = 0
        self.num_samples = 0

    def close(self):
        if getattr(self, "_h", None):
            self._lib.SLAEncoder_Destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, rc, where):
        if rc != 0:
            raise SlaError(rc, where)

    def set_option(self, name, value):
        """sla_hip_encoder_set_option: layout knobs and route switches of this handle (include/sla_hip.h)"""
        self._check(self._lib.sla_hip_encoder_set_option(self._h, name.encode(), float(value)), "sla_hip_encoder_set_option(%s)" % name)

    def set_wave_format(self, num_channels, bit_per_sample, sampling_rate, offset_lshift=0):
        wf = SLAWaveFormat(num_channels, bit_per_sample, sampling_rate, offset_lshift)
        self._check(self._lib.SLAEncoder_SetWaveFormat(self._h, C.byref(wf)), "SLAEncoder_SetWaveFormat")
        self.num_channels = num_channels

    def set_encode_parameter(self, parcor_order, longterm_order, lms_order_per_filter,
                             ch_process_method=CH_NONE, window_function_type=WINDOW_SIN,
                             max_num_block_samples=4096):
        ep = SLAEncodeParameter(parcor_order, longterm_order, lms_order_per_filter, ch_process_method,
                                window_function_type, max_num_block_samples)
        self._check(self._lib.SLAEncoder_SetEncodeParameter(self._h, C.byref(ep)), "SLAEncoder_SetEncodeParameter")
        self.order, self.ltm_order = parcor_order, longterm_order

    @staticmethod
    def _planes(pcm):
        pcm = np.ascontiguousarray(pcm, np.int32)
        ptrs = (i32p * pcm.shape[0])(*[pcm[c].ctypes.data_as(i32p) for c in range(pcm.shape[0])])
        return pcm, ptrs

    def encode_whole(self, pcm, capacity=None, out=None):
        """planar left-justified int32 [C][N] on the host -> .sla bytes.  `out`: optional preallocated
        uint8 array to encode into (a view of it is returned instead of a bytes copy)"""
        pcm, ptrs = self._planes(pcm)
        n = pcm.shape[1]
        if out is None:
            cap = capacity if capacity is not None else 8 * pcm.shape[0] * n + 65536
            buf = np.zeros(cap, np.uint8)
        else:
            buf, cap = out, min(len(out), 0xFFFFFFF0)        # the API's sizes are 32-bit
        size = C.c_uint32(0)
        self._check(self._lib.SLAEncoder_EncodeWhole(self._h, ptrs, n, buf.ctypes.data_as(u8p), cap, C.byref(size)),
                    "SLAEncoder_EncodeWhole")
        self.num_samples = n
        return buf[:size.value] if out is not None else buf[:size.value].tobytes()

    def encode_batch(self, pcms, capacities=None, outs=None):
        """many files in one pass (sla_hip_encode_batch): list of planar int32 [C][n_i] -> list of (result, bytes).
        `outs`: optional preallocated uint8 arrays, one per file (views of them are returned instead of copies)"""
        keep, items = [], (BatchItem * len(pcms))()
        for i, pcm in enumerate(pcms):
            pcm, ptrs = self._planes(pcm)
            if outs is not None:
                buf, cap = outs[i], len(outs[i])
            else:
                cap = capacities[i] if capacities is not None else 8 * pcm.shape[0] * pcm.shape[1] + 65536
                buf = np.empty(cap, np.uint8)
            keep.append((pcm, ptrs, buf))
            items[i].input = ptrs
            items[i].num_samples = pcm.shape[1]
            items[i].data = buf.ctypes.data_as(u8p)
            items[i].data_size = cap
        self._check(self._lib.sla_hip_encode_batch(self._h, items, len(pcms)), "sla_hip_encode_batch")
        if outs is not None:
            return [(int(items[i].result), keep[i][2][:items[i].output_size]) for i in range(len(pcms))]
        return [(int(items[i].result), keep[i][2][:items[i].output_size].tobytes()) for i in range(len(pcms))]

    def encode_block(self, pcm, capacity=None):
        pcm, ptrs = self._planes(pcm)
        n = pcm.shape[1]
        cap = capacity if capacity is not None else 8 * pcm.shape[0] * n + 4096
        out = np.zeros(cap, np.uint8)
        size = C.c_uint32(0)
        self._check(self._lib.SLAEncoder_EncodeBlock(self._h, ptrs, n, out.ctypes.data_as(u8p), cap, C.byref(size)),
                    "SLAEncoder_EncodeBlock")
        self.num_samples = n
        return out[:size.value].tobytes()

    def analyze_device(self, device_ptr, plane_stride, num_samples, stream=None):
        """hot path on PCM already resident in HBM; returns the 12 stage timings [ms]"""
        timing = (C.c_float * 12)()
        self._check(self._lib.sla_hip_analyze_device(self._h, C.c_void_p(device_ptr), plane_stride, num_samples,
                                                     C.c_void_p(stream) if stream else None, timing),
                    "sla_hip_analyze_device")
        self.num_samples = num_samples
        return list(timing)

    def analyze_batch_device(self, device_ptr, plane_stride, span, starts, lens):
        """hot path on a batch of files resident in HBM (files at starts[i], multiples of 1024); returns
        (the 12 stage timings [ms], offset_lshift per file)"""
        st = np.ascontiguousarray(starts, np.uint32)
        ln = np.ascontiguousarray(lens, np.uint32)
        lsh = np.zeros(len(st), np.uint32)
        timing = (C.c_float * 12)()
        self._check(self._lib.sla_hip_analyze_batch_device(self._h, C.c_void_p(device_ptr), plane_stride, span,
                                                           st.ctypes.data_as(u32p), ln.ctypes.data_as(u32p), len(st),
                                                           lsh.ctypes.data_as(u32p), timing), "sla_hip_analyze_batch_device")
        self.num_samples = span
        return list(timing), lsh

    def shard_scan(self, device_ptr, plane_stride, num_samples):
        """sla_hip_shard_scan: (OR of every sample word, 1-bit 'not silent' mask as uint64 words) of a piece of a file"""
        orw = C.c_uint32(0)
        mask = np.zeros((num_samples + 63) // 64, np.uint64)
        self._check(self._lib.sla_hip_shard_scan(self._h, C.c_void_p(device_ptr), plane_stride, num_samples, C.byref(orw),
                                                 mask.ctypes.data_as(C.POINTER(C.c_uint64))), "sla_hip_shard_scan")
        return int(orw.value), mask

    def shard_scan_counts(self, device_ptr, plane_stride, num_samples):
        """sla_hip_shard_scan_counts: (OR of every sample word, number of all-zero 64-sample mask words) of a piece"""
        orw, zw = C.c_uint32(0), C.c_uint32(0)
        self._check(self._lib.sla_hip_shard_scan_counts(self._h, C.c_void_p(device_ptr), plane_stride, num_samples,
                                                        C.byref(orw), C.byref(zw)), "sla_hip_shard_scan_counts")
        return int(orw.value), int(zw.value)

    def shard_analyze(self, device_ptr, plane_stride, num_samples, file_or_word, no_silence=False):
        """sla_hip_shard_analyze: the hot path on a range of a longer file whose OR word is `file_or_word`
        (no_silence: sla_hip_shard_analyze_no_silence -- the file's scan counted no all-zero mask word)"""
        timing = (C.c_float * 12)()
        fn = self._lib.sla_hip_shard_analyze_no_silence if (no_silence and file_or_word != 0) else self._lib.sla_hip_shard_analyze
        self._check(fn(self._h, C.c_void_p(device_ptr), plane_stride, num_samples,
                       C.c_uint32(file_or_word), timing), "sla_hip_shard_analyze")
        self.num_samples = num_samples
        return list(timing)

    def last_timing(self):
        """the 12 stage timings / counters of the last analysis (see include/sla_hip.h)"""
        timing = (C.c_float * 12)()
        self._check(self._lib.sla_hip_last_timing(self._h, timing), "sla_hip_last_timing")
        return list(timing)

    def last_kernel_ms(self):
        """on-device execution time [ms] of (k_lpc_blocks, k_lattice, k_ltm_acf, k_tail) in the last analysis"""
        k = (C.c_float * 4)()
        self._check(self._lib.sla_hip_last_kernel_ms(self._h, k), "sla_hip_last_kernel_ms")
        return list(k)

    def last_counters(self):
        """(search groups rerun as chains, super-frames planned on the host, exact search used, device plan enabled,
        k_tail launches, long-term solve on the device)"""
        c = (C.c_uint32 * 6)()
        self._check(self._lib.sla_hip_last_counters(self._h, c), "sla_hip_last_counters")
        return tuple(c)

    def last_expand(self):
        """(pipeline chunks whose block stage was launched from device-written tables, pipeline chunks, analyses of this
        handle served from kept search tables, searches launched on a wrong guess of the prepass result)"""
        c = (C.c_uint32 * 4)()
        self._check(self._lib.sla_hip_last_expand(self._h, c), "sla_hip_last_expand")
        return tuple(c)

    def last_block_cert(self):
        """(1 if the block stage took the certified route, (block, channel) pairs redone by the exact kernels)"""
        c = (C.c_uint32 * 2)()
        self._check(self._lib.sla_hip_last_block_cert(self._h, c), "sla_hip_last_block_cert")
        return tuple(c)

    def last_cert_audit(self):
        """option cert_audit: (certified pairs the exact kernels re-analysed and found equal, pairs they found different)"""
        c = (C.c_uint32 * 2)()
        self._check(self._lib.sla_hip_last_cert_audit(self._h, c), "sla_hip_last_cert_audit")
        return tuple(c)

    def bind_residual_planes(self, lattice_ptr, final_ptr, plane_stride):
        self._check(self._lib.sla_hip_bind_residual_planes(self._h, C.c_void_p(lattice_ptr), C.c_void_p(final_ptr),
                                                           plane_stride), "sla_hip_bind_residual_planes")

    def pack(self, capacity, on_device=False):
        """bit-pack the analysed file: host threads (sla_hip_pack) or device kernels (sla_hip_pack_device)"""
        out = np.zeros(capacity, np.uint8)
        size = C.c_uint32(0)
        fn = self._lib.sla_hip_pack_device if on_device else self._lib.sla_hip_pack
        self._check(fn(self._h, out.ctypes.data_as(u8p), capacity, C.byref(size)), "sla_hip_pack")
        return out[:size.value].tobytes()

    def trace(self, want_residuals=True, max_blocks=None):
        tr = Trace(self.num_channels, self.order, self.ltm_order, self.num_samples,
                   max_blocks if max_blocks is not None else self.num_samples // 1024 + 8, want_residuals)
        self._check(self._lib.sla_hip_get_trace(self._h, C.byref(tr.c)), "sla_hip_get_trace")
        return tr

    def final_residual_ptr(self):
        stride = C.c_uint64(0)
        return self._lib.sla_hip_final_residual(self._h, C.byref(stride)), stride.value


def shard_bounds(num_samples, max_num_block_samples, nz_mask, world):
    """sla_hip_shard_bounds: [bounds[r], bounds[r+1]) = the samples rank r of `world` encodes, every bound a super-frame
    start of the whole file's hop over silence runs.  Pure host arithmetic of the library (no GPU needed)."""
    bounds = np.zeros(world + 1, np.uint32)
    if nz_mask is None:            # no rank counted an all-zero mask word (shard_scan_counts): nothing is silent
        ptr = None
    else:
        mask = np.ascontiguousarray(nz_mask, np.uint64)
        if len(mask) < (num_samples + 63) // 64:
            raise ValueError("mask too short")
        ptr = mask.ctypes.data_as(C.POINTER(C.c_uint64))
    rc = lib().sla_hip_shard_bounds(C.c_uint32(num_samples), C.c_uint32(max_num_block_samples),
                                    ptr, C.c_uint32(world), bounds.ctypes.data_as(u32p))
    if rc != 0:
        raise SlaError(rc, "sla_hip_shard_bounds")
    return [int(b) for b in bounds]


def shard_join(shard_images):
    """the per-rank .sla images (43-byte header + blocks each, in rank order) -> the file: sla_hip_shard_header over
    the shards' headers, then the blocks back to back"""
    shard_images = [img for img in shard_images if len(img) != 0]        # a rank without super-frames contributes nothing
    if not shard_images:
        raise ValueError("no shard produced an image")
    heads = [np.frombuffer(bytes(img[:43]), np.uint8).copy() for img in shard_images]
    if any(len(h) != 43 for h in heads):
        raise ValueError("shard image shorter than a header")
    ptrs = (u8p * len(heads))(*[h.ctypes.data_as(u8p) for h in heads])
    out = np.zeros(43, np.uint8)
    rc = lib().sla_hip_shard_header(ptrs, len(heads), out.ctypes.data_as(u8p), 43)
    if rc != 0:
        raise SlaError(rc, "sla_hip_shard_header")
    return out.tobytes() + b"".join(bytes(img[43:]) for img in shard_images)


class Decoder:
    """Python face of ``struct SLADecoder`` (reference src/include/public/SLADecoder.h)."""

    def __init__(self, max_num_channels=8, max_num_block_samples=16384, max_parcor_order=48,
                 max_longterm_order=5, max_lms_order_per_filter=32, enable_crc_check=1):
        self._lib = lib()
        cfg = SLADecoderConfig(max_num_channels, max_num_block_samples, max_parcor_order,
                               max_longterm_order, max_lms_order_per_filter, enable_crc_check, 0)
        self._h = self._lib.SLADecoder_Create(C.byref(cfg))
        if not self._h:
            raise RuntimeError("SLADecoder_Create failed: no usable HIP device or unsupported capacity "
                               "(libsla_hip has no CPU fallback)")

    def close(self):
        if getattr(self, "_h", None):
            self._lib.SLADecoder_Destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def decode_whole(self, data, capacity, num_channels=None):
        """.sla bytes -> (result code, planar left-justified int32 [C][n]); like the reference, samples of the
        blocks before a failing one are returned together with the error code"""
        buf = np.frombuffer(bytes(data), np.uint8)
        nch = num_channels if num_channels is not None else (int(buf[14]) if len(buf) > 14 else 1)
        out = np.zeros((max(nch, 1), max(capacity, 1)), np.int32)
        ptrs = (i32p * out.shape[0])(*[out[c].ctypes.data_as(i32p) for c in range(out.shape[0])])
        n = C.c_uint32(0)
        rc = self._lib.SLADecoder_DecodeWhole(self._h, buf.ctypes.data_as(u8p), len(buf), ptrs, capacity, C.byref(n))
        return rc, out[:, :n.value]

    def decode_device(self, data, image_ptr, planes_ptr, plane_stride):
        """decode an image that already lives in device memory into device planes; returns (rc, samples)"""
        buf = np.frombuffer(bytes(data), np.uint8) if not isinstance(data, np.ndarray) else data
        n = C.c_uint32(0)
        rc = self._lib.sla_hip_decode_device(self._h, buf.ctypes.data_as(u8p), C.c_void_p(image_ptr), len(buf),
                                             C.c_void_p(planes_ptr), plane_stride, C.byref(n))
        return rc, n.value

    def last_timing(self):
        """[ms] upload, block walk, kernels (stream events), download, total; number of kernel batches"""
        t = (C.c_float * 6)()
        self._lib.sla_hip_decoder_last_timing(self._h, t)
        return list(t)


def streaming_decode(data, decode_interval_hz=120.0, feed=None, max_bit_per_sample=24, crc=1,
                     capacity=(8, 16384, 48, 5, 40)):
    """Drive SLAStreamingDecoder_* the way the reference CLI does (src/main.c:277-420): header, then per call
    append the estimated number of bytes (or `feed(i)` bytes), decode, collect.  Returns (rc, pcm [C][n], calls)."""
    L = lib()
    buf = np.frombuffer(bytes(data), np.uint8)
    h = SLAHeaderInfo()
    rc = L.SLADecoder_DecodeHeader(buf.ctypes.data_as(u8p), len(buf), C.byref(h))
    if rc != 0:
        return rc, None, 0
    cfg = SLAStreamingDecoderConfig(SLADecoderConfig(*capacity, crc, 0), decode_interval_hz, max_bit_per_sample)
    dec = L.SLAStreamingDecoder_Create(C.byref(cfg))
    if not dec:
        raise RuntimeError("SLAStreamingDecoder_Create failed")
    try:
        rc = L.SLAStreamingDecoder_SetWaveFormat(dec, C.byref(h.wave_format))
        if rc == 0:
            rc = L.SLAStreamingDecoder_SetEncodeParameter(dec, C.byref(h.encode_param))
        if rc != 0:
            return rc, None, 0
        nch, total = h.wave_format.num_channels, h.num_samples
        out = np.zeros((nch, max(total, 1)), np.int32)
        sample_pos, data_pos, calls = 0, 43, 0
        est, got = C.c_uint32(0), C.c_uint32(0)
        dummy_p, dummy_n = u8p(), C.c_uint32(0)
        while sample_pos < total:
            if feed is not None:
                want = int(feed(calls))
            elif sample_pos == 0 and calls == 0:
                want = h.max_block_size
            else:
                L.SLAStreamingDecoder_EstimateMinimumNessesaryDataSize(dec, C.byref(est))
                want = est.value
            put = min(want, len(buf) - data_pos)
            rc = L.SLAStreamingDecoder_AppendDataFragment(dec, buf[data_pos:].ctypes.data_as(u8p) if put > 0 else buf.ctypes.data_as(u8p), put)
            if rc != 0:
                return rc, out[:, :sample_pos], calls
            ptrs = (i32p * nch)(*[out[c, sample_pos:].ctypes.data_as(i32p) for c in range(nch)])
            rc = L.SLAStreamingDecoder_Decode(dec, ptrs, total - sample_pos, C.byref(got))
            calls += 1
            if rc == 9 and data_pos + put < len(buf):         # not even a block header yet: keep feeding
                rc, got.value = 0, 0
            if rc != 0:
                return rc, out[:, :sample_pos], calls
            L.SLAStreamingDecoder_CollectDataFragment(dec, C.byref(dummy_p), C.byref(dummy_n))
            data_pos += put
            sample_pos += got.value
            if got.value == 0 and put == 0 and data_pos >= len(buf):
                return 9, out[:, :sample_pos], calls          # the stream ended inside a block
        return 0, out[:, :total], calls
    finally:
        L.SLAStreamingDecoder_Destroy(dec)


def decode_header(data):
    """(result code, SLAHeaderInfo) of the first 43 bytes"""
    buf = np.frombuffer(bytes(data), np.uint8)
    h = SLAHeaderInfo()
    rc = lib().SLADecoder_DecodeHeader(buf.ctypes.data_as(u8p), len(buf), C.byref(h))
    return rc, h


def encode_header(num_channels, bits, rate, lshift, parcor, ltm, lms, chproc, window, max_block,
                  num_samples, num_blocks, max_block_size, max_bps):
    h = SLAHeaderInfo(SLAWaveFormat(num_channels, bits, rate, lshift),
                      SLAEncodeParameter(parcor, ltm, lms, chproc, window, max_block),
                      num_samples, num_blocks, max_block_size, max_bps)
    out = np.zeros(64, np.uint8)
    rc = lib().SLAEncoder_EncodeHeader(C.byref(h), out.ctypes.data_as(u8p), 64)
    if rc != 0:
        raise SlaError(rc, "SLAEncoder_EncodeHeader")
    return out[:43].tobytes()


def device_name():
    return lib().sla_hip_device_name().decode()
