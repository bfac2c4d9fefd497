"""Input validation in front of the hot path (matchering/checker.py:90-142): length limits, mono ->
stereo, resampling to ``config.internal_sample_rate``, clipping / limiter detection on the target,
and the "target equals reference" guard.  Host numpy: none of this is on the timed path.

Resampling: the reference calls ``resampy.resample`` (Kaiser-windowed sinc).  ``resampy`` is used
when it is importable; otherwise ``scipy.signal.resample_poly`` (polyphase Kaiser FIR) does the
rate change -- same purpose, not sample-identical to resampy, and only reached when a file's rate
differs from the internal rate.
"""

from math import gcd

import numpy as np

from .config import Config
from .log import Code, ModuleError, debug, info, warning
from .utils import time_str


def count_max_peaks(array: np.ndarray):
    """dsp.py:49-54: the peak magnitude and how many samples sit on it (numpy.isclose)."""
    max_value = np.abs(array).max()
    hits = np.isclose(array, max_value) | np.isclose(array, -max_value)
    return max_value, int(np.count_nonzero(hits))


def _resample(array, sample_rate, required):
    try:                                            # pragma: no cover - not installed in the build image
        from resampy import resample

        return resample(array, sample_rate, required, axis=0)
    except ImportError:
        from scipy.signal import resample_poly

        g = gcd(int(required), int(sample_rate))
        return resample_poly(array, int(required) // g, int(sample_rate) // g, axis=0)


def check(array: np.ndarray, sample_rate: int, config: Config, name: str):
    """checker.py:90-137: returns the validated ``(array (n, 2), internal_sample_rate)``."""
    name = name.upper()
    target = name == "TARGET"
    length = array.shape[0]
    debug(f"{name}: {length} frames = {time_str(length, sample_rate)}")
    if length > config.max_length * sample_rate:
        raise ModuleError(Code.ERROR_TARGET_LENGTH_IS_EXCEEDED if target
                          else Code.ERROR_REFERENCE_LENGTH_LENGTH_IS_EXCEEDED)
    if length < config.fft_size * sample_rate // config.internal_sample_rate:
        raise ModuleError(Code.ERROR_TARGET_LENGTH_IS_TOO_SMALL if target
                          else Code.ERROR_REFERENCE_LENGTH_LENGTH_TOO_SMALL)

    if array.shape[1] == 1:
        info(Code.INFO_TARGET_IS_MONO if target else Code.INFO_REFERENCE_IS_MONO)
        array = np.repeat(array, repeats=2, axis=1)                     # dsp.py:45-46
    elif array.shape[1] != 2:
        raise ModuleError(Code.ERROR_TARGET_NUM_OF_CHANNELS_IS_EXCEEDED if target
                          else Code.ERROR_REFERENCE_NUM_OF_CHANNELS_IS_EXCEEDED)

    if sample_rate != config.internal_sample_rate:
        debug(f"{name}: converting {sample_rate} Hz -> {config.internal_sample_rate} Hz")
        array = _resample(array, sample_rate, config.internal_sample_rate)
        if target:
            warning(Code.WARNING_TARGET_IS_RESAMPLED)
        else:
            info(Code.INFO_REFERENCE_IS_RESAMPLED)
        sample_rate = config.internal_sample_rate

    if target:
        max_value, max_count = count_max_peaks(array)
        if max_count > config.clipping_samples_threshold:
            if np.isclose(max_value, 1.0):
                warning(Code.WARNING_TARGET_IS_CLIPPING)
            elif max_count > config.limited_samples_threshold:
                warning(Code.WARNING_TARGET_LIMITER_IS_APPLIED)
    return array, sample_rate


def check_equality(target: np.ndarray, reference: np.ndarray) -> None:
    """checker.py:140-142."""
    if target.shape == reference.shape and np.allclose(target, reference):
        raise ModuleError(Code.ERROR_TARGET_EQUALS_REFERENCE)
