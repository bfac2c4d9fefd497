"""ctypes binding of libmgx.so (include/mgx.h).

The library is the product: there is no Python or numpy implementation of the
hot path behind it.  If the shared object is missing it is built in place with
hipcc (matchering_amd/build.py); if that is impossible the import fails loudly.
"""

import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("MGX_LIB") or os.path.join(_HERE, "libmgx.so")

c_float_p = ctypes.POINTER(ctypes.c_float)
c_double_p = ctypes.POINTER(ctypes.c_double)
c_int32_p = ctypes.POINTER(ctypes.c_int32)
c_int64_p = ctypes.POINTER(ctypes.c_int64)


class MgxConfig(ctypes.Structure):
    """mgx_config (include/mgx.h)."""

    _fields_ = [
        ("internal_sample_rate", ctypes.c_int32),
        ("fft_size", ctypes.c_int32),
        ("lin_log_oversampling", ctypes.c_int32),
        ("rms_correction_steps", ctypes.c_int32),
        ("max_piece_size", ctypes.c_double),
        ("threshold", ctypes.c_double),
        ("min_value", ctypes.c_double),
        ("lowess_frac", ctypes.c_double),
        ("lowess_it", ctypes.c_int32),
        ("reserved0", ctypes.c_int32),
        ("lowess_delta", ctypes.c_double),
        ("attack_ms", ctypes.c_double),
        ("hold_ms", ctypes.c_double),
        ("release_ms", ctypes.c_double),
        ("attack_filter_coefficient", ctypes.c_double),
        ("hold_filter_order", ctypes.c_int32),
        ("release_filter_order", ctypes.c_int32),
        ("hold_filter_coefficient", ctypes.c_double),
        ("release_filter_coefficient", ctypes.c_double),
    ]


class MgxReport(ctypes.Structure):
    """mgx_report (include/mgx.h)."""

    _fields_ = [
        ("final_amplitude_coefficient", ctypes.c_double),
        ("target_match_rms", ctypes.c_double),
        ("reference_match_rms", ctypes.c_double),
        ("rms_coefficient", ctypes.c_double),
        ("correction_coefficients", ctypes.c_double * 16),
        ("normalize_coefficient", ctypes.c_double),
        ("result_peak", ctypes.c_double),
        ("target_divisions", ctypes.c_int32),
        ("reference_divisions", ctypes.c_int32),
        ("target_piece", ctypes.c_int64),
        ("reference_piece", ctypes.c_int64),
        ("target_loud_count", ctypes.c_int32),
        ("reference_loud_count", ctypes.c_int32),
        ("limiter_active", ctypes.c_int32),
        ("reserved", ctypes.c_int32),
    ]


# every symbol include/mgx.h declares: name -> (restype, argtypes)
_VP = ctypes.c_void_p
SYMBOLS = {
    "mgx_version": (ctypes.c_int, []),
    "mgx_last_error": (ctypes.c_char_p, []),
    "mgx_device_count": (ctypes.c_int, [ctypes.POINTER(ctypes.c_int)]),
    "mgx_device_pci_bus_id": (ctypes.c_int, [ctypes.c_int, ctypes.c_char_p, ctypes.c_int32]),
    "mgx_create": (ctypes.c_int, [ctypes.c_int, ctypes.POINTER(_VP)]),
    "mgx_destroy": (ctypes.c_int, [_VP]),
    "mgx_config_default": (ctypes.c_int, [ctypes.POINTER(MgxConfig)]),
    "mgx_malloc": (ctypes.c_int, [_VP, ctypes.c_size_t, ctypes.POINTER(_VP)]),
    "mgx_free": (ctypes.c_int, [_VP, _VP]),
    "mgx_memcpy_h2d": (ctypes.c_int, [_VP, _VP, _VP, ctypes.c_size_t]),
    "mgx_memcpy_d2h": (ctypes.c_int, [_VP, _VP, _VP, ctypes.c_size_t]),
    "mgx_synchronize": (ctypes.c_int, [_VP]),
    "mgx_timer_start": (ctypes.c_int, [_VP]),
    "mgx_timer_stop": (ctypes.c_int, [_VP, c_float_p]),
    "mgx_master": (ctypes.c_int, [_VP, _VP, ctypes.c_int64, _VP, ctypes.c_int64, ctypes.POINTER(MgxConfig),
                                  _VP, _VP, _VP, ctypes.POINTER(MgxReport)]),
    "mgx_master_with_fir": (ctypes.c_int, [_VP, _VP, ctypes.c_int64, _VP, ctypes.c_int64, ctypes.POINTER(MgxConfig),
                                           _VP, _VP, _VP, _VP, ctypes.POINTER(MgxReport)]),
    "mgx_analyze": (ctypes.c_int, [_VP, _VP, ctypes.c_int64, ctypes.POINTER(MgxConfig), ctypes.c_int,
                                   c_double_p, c_double_p, c_double_p, c_int32_p, c_int64_p,
                                   c_double_p, c_int32_p, c_double_p, c_double_p]),
    "mgx_design_fir": (ctypes.c_int, [ctypes.POINTER(MgxConfig), c_double_p, c_double_p, c_double_p,
                                      c_double_p, c_double_p]),
    "mgx_convolve": (ctypes.c_int, [_VP, _VP, ctypes.c_int64, c_double_p, c_double_p, ctypes.c_int32,
                                    ctypes.c_double, _VP, _VP, c_double_p]),
    "mgx_host_alloc": (ctypes.c_int, [ctypes.c_size_t, ctypes.POINTER(_VP)]),
    "mgx_host_free": (ctypes.c_int, [_VP]),
    "mgx_memcpy_h2d_async": (ctypes.c_int, [_VP, _VP, _VP, ctypes.c_size_t]),
    "mgx_memcpy_d2h_async": (ctypes.c_int, [_VP, _VP, _VP, ctypes.c_size_t]),
    "mgx_stage_timing": (ctypes.c_int, [_VP, ctypes.c_int32]),
    "mgx_stage_times": (ctypes.c_int, [_VP, c_float_p]),
    "mgx_code_bytes": (ctypes.c_int, [c_int32_p, ctypes.c_int32]),
    "mgx_clipped_piece_sumsq": (ctypes.c_int, [_VP, _VP, ctypes.c_int64, ctypes.c_int64, ctypes.c_int32,
                                               ctypes.c_double, c_double_p]),
    "mgx_limit": (ctypes.c_int, [_VP, _VP, ctypes.c_int64, ctypes.POINTER(MgxConfig), ctypes.c_double,
                                 ctypes.c_double, _VP, c_int32_p]),
    "mgx_scale": (ctypes.c_int, [_VP, _VP, ctypes.c_int64, ctypes.c_double, _VP]),
    "mgx_peak_count": (ctypes.c_int, [_VP, _VP, ctypes.c_int64, c_double_p, ctypes.POINTER(ctypes.c_int64)]),
    "mgx_pcm_decode": (ctypes.c_int, [_VP, _VP, ctypes.c_int64, ctypes.c_int32, _VP]),
    "mgx_pcm_encode": (ctypes.c_int, [_VP, _VP, ctypes.c_int64, ctypes.c_int32, _VP]),
    "mgx_window_energy": (ctypes.c_int, [_VP, _VP, ctypes.c_int64, ctypes.c_int64, ctypes.c_int64, c_double_p,
                                         ctypes.c_int64, c_int64_p]),
    "mgx_preview_cut": (ctypes.c_int, [_VP, _VP, ctypes.c_int64, ctypes.c_int64, ctypes.c_int64, ctypes.c_int64,
                                       ctypes.c_double, _VP]),
    "mgx_last_fir": (ctypes.c_int, [_VP, ctypes.POINTER(_VP), c_int32_p]),
    "mgx_comm_unique_id": (ctypes.c_int, [_VP]),
    "mgx_comm_init": (ctypes.c_int, [_VP, _VP, ctypes.c_int, ctypes.c_int]),
    "mgx_comm_count": (ctypes.c_int, [_VP, c_int32_p]),
    "mgx_comm_broadcast_f32": (ctypes.c_int, [_VP, _VP, ctypes.c_int64, ctypes.c_int]),
    "mgx_comm_allgather_f32": (ctypes.c_int, [_VP, _VP, _VP, ctypes.c_int64]),
    "mgx_comm_destroy": (ctypes.c_int, [_VP]),
}


# enum mgx_stage (include/mgx.h), in order
STAGES = ("analyze", "design_fir", "filter_spectra", "convolve", "correct_levels", "scale_outputs", "limit")


ERR_RETRY = -6          # enum mgx_status MGX_ERR_RETRY (include/mgx.h)


class MgxError(RuntimeError):
    """A libmgx call failed; ``code`` is the negative mgx_status."""

    def __init__(self, code, message):
        super().__init__(f"libmgx error {code}: {message}")
        self.code = code

    @property
    def retry(self):
        """The handle recovered from a device-side wait that expired (the GPU is shared): what was queued since the
        last synchronisation is lost, the same calls made again succeed (``MGX_ERR_RETRY``)."""
        return self.code == ERR_RETRY


_lib = None


def library():
    """Load (building first if needed) libmgx.so and declare every prototype."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.environ.get("MGX_LIB"):
        # build() compares a digest of csrc/ and include/mgx.h with the one the binary was made from
        # and returns at once when they agree: an edited kernel or struct never meets a stale binary
        from . import build as _build

        try:
            _build.build()
        except Exception as exc:               # noqa: BLE001 -- no hipcc on this host, a read-only tree ...
            if not os.path.exists(LIB_PATH):
                raise
            import warnings

            warnings.warn(f"libmgx.so could not be rebuilt from the sources beside it ({exc!r}); loading the "
                          f"existing binary, which may be older than they are", RuntimeWarning)
    lib = ctypes.CDLL(LIB_PATH)
    for name, (restype, argtypes) in SYMBOLS.items():
        fn = getattr(lib, name)          # AttributeError here = header and library disagree
        fn.restype = restype
        fn.argtypes = argtypes
    _lib = lib
    return lib


def check(code):
    if code != 0:
        raise MgxError(code, library().mgx_last_error().decode("utf-8", "replace"))
