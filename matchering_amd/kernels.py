"""Stage-level entry points of libmgx on numpy arrays (upload -> kernel -> download).

These are the finer boundaries of SURVEY.md section 8(b): each wraps one C-ABI call so
that a stage can be used, and parity-tested, on its own.  Every wrapper holds the
device's lock from its upload to its download (include/mgx.h: one caller per handle).
"""

import functools

import ctypes
from collections import namedtuple

import numpy as np

from ._native import c_double_p, check, library
from .device import default_device

LevelStats = namedtuple("LevelStats", "peak amplitude_coefficient match_rms divisions piece_size "
                                      "rmses loud average_spectrum_mid average_spectrum_side")


def _dp(a):
    return a.ctypes.data_as(c_double_p)


def _exclusive(fn):
    """Run ``fn`` under the lock of the device it will use (``device=`` keyword or the default one)."""

    @functools.wraps(fn)
    def locked(*args, device=None, **kwargs):
        dev = device or default_device()
        with dev.lock:
            return fn(*args, device=dev, **kwargs)

    return locked


@_exclusive
def analyze(array, config, is_reference=False, device=None):
    """match_levels.py:134-161 + match_frequencies.py:30-42 in one pass (``mgx_analyze``)."""
    dev = device or default_device()
    x = np.ascontiguousarray(array, dtype=np.float32)
    n = x.shape[0]
    native = config.to_native()
    max_div = int(n / config.max_piece_size) + 1
    half = config.fft_size // 2
    peak, amp, match = ctypes.c_double(), ctypes.c_double(), ctypes.c_double()
    div, piece = ctypes.c_int32(), ctypes.c_int64()
    rms = np.zeros(max_div)
    loud = np.zeros(max_div, dtype=np.int32)
    avg_mid, avg_side = np.zeros(half + 1), np.zeros(half + 1)
    buf = dev.upload(x)
    try:
        check(library().mgx_analyze(
            dev.handle, ctypes.c_void_p(buf.ptr), n, ctypes.byref(native), int(bool(is_reference)),
            ctypes.byref(peak), ctypes.byref(amp), ctypes.byref(match), ctypes.byref(div),
            ctypes.byref(piece), _dp(rms), loud.ctypes.data_as(ctypes.POINTER(ctypes.c_int32)),
            _dp(avg_mid), _dp(avg_side)))
    finally:
        buf.release()
    d = div.value
    return LevelStats(peak.value, amp.value, match.value, d, piece.value, rms[:d].copy(),
                      loud[:d].astype(bool), avg_mid, avg_side)


def design_fir(avg_target, avg_reference, config):
    """match_frequencies.py:78-101 from averaged spectra (``mgx_design_fir``, host float64).
    Returns (taps, raw curve, smoothed curve)."""
    native = config.to_native()
    half = config.fft_size // 2
    a_t = np.ascontiguousarray(avg_target, dtype=np.float64)
    a_r = np.ascontiguousarray(avg_reference, dtype=np.float64)
    taps, raw, smooth = np.zeros(config.fft_size), np.zeros(half + 1), np.zeros(half + 1)
    check(library().mgx_design_fir(ctypes.byref(native), _dp(a_t), _dp(a_r), _dp(taps), _dp(raw), _dp(smooth)))
    return taps, raw, smooth


@_exclusive
def convolve(array, mid_fir, side_fir, gain=1.0, device=None):
    """match_frequencies.py:104-119 on interleaved L/R frames (``mgx_convolve``).
    Returns (result (n,2) float32, result_mid (n,) float32, peak)."""
    dev = device or default_device()
    x = np.ascontiguousarray(array, dtype=np.float32)
    n = x.shape[0]
    hm = np.ascontiguousarray(mid_fir, dtype=np.float64)
    hs = np.ascontiguousarray(side_fir, dtype=np.float64)
    assert hm.shape == hs.shape and hm.ndim == 1
    xb, yb, mb = dev.upload(x), dev.alloc(n * 8), dev.alloc(n * 4)
    peak = ctypes.c_double()
    try:
        check(library().mgx_convolve(dev.handle, ctypes.c_void_p(xb.ptr), n, _dp(hm), _dp(hs), hm.shape[0],
                                     float(gain), ctypes.c_void_p(yb.ptr), ctypes.c_void_p(mb.ptr),
                                     ctypes.byref(peak)))
        return dev.download(yb, (n, 2)), dev.download(mb, (n,)), peak.value
    finally:
        for b in (xb, yb, mb):
            b.release()


@_exclusive
def clipped_piece_sumsq(mid, piece_size, divisions, gain=1.0, device=None):
    """Sum of squares per piece of clip(gain*mid, -1, 1) (stages.py:149-160, ``mgx_clipped_piece_sumsq``)."""
    dev = device or default_device()
    m = np.ascontiguousarray(mid, dtype=np.float32)
    out = np.zeros(divisions)
    buf = dev.upload(m)
    try:
        check(library().mgx_clipped_piece_sumsq(dev.handle, ctypes.c_void_p(buf.ptr), m.shape[0],
                                                int(piece_size), int(divisions), float(gain), _dp(out)))
    finally:
        buf.release()
    return out


@_exclusive
def limit(array, config, gain=1.0, post_gain=1.0, device=None):
    """limiter/hyrax.py:78-99 on (array*gain), times post_gain (``mgx_limit``).
    Returns (limited (n,2) float32, active flag)."""
    dev = device or default_device()
    x = np.ascontiguousarray(array, dtype=np.float32)
    n = x.shape[0]
    native = config.to_native()
    active = ctypes.c_int32()
    xb, ob = dev.upload(x), dev.alloc(n * 8)
    try:
        check(library().mgx_limit(dev.handle, ctypes.c_void_p(xb.ptr), n, ctypes.byref(native), float(gain),
                                  float(post_gain), ctypes.c_void_p(ob.ptr), ctypes.byref(active)))
        return dev.download(ob, (n, 2)), bool(active.value)
    finally:
        xb.release()
        ob.release()


@_exclusive
def scale(array, gain, device=None):
    """dsp.py:89-90 ``amplify`` on the device (``mgx_scale``)."""
    dev = device or default_device()
    x = np.ascontiguousarray(array, dtype=np.float32)
    n = x.shape[0]
    xb, ob = dev.upload(x), dev.alloc(n * 8)
    try:
        check(library().mgx_scale(dev.handle, ctypes.c_void_p(xb.ptr), n, float(gain), ctypes.c_void_p(ob.ptr)))
        dev.synchronize()
        return dev.download(ob, (n, 2))
    finally:
        xb.release()
        ob.release()
