"""Input validation in front of the hot path (matchering/checker.py:90-142): length limits, mono ->
stereo, resampling to ``config.internal_sample_rate``, clipping / limiter detection on the target,
and the "target equals reference" guard.  Host numpy: none of this is on the timed path.

Resampling: the reference calls ``resampy.resample`` (Kaiser-windowed sinc table, ``kaiser_best``).  ``resampy`` is
used when it is importable; otherwise ``matchering_amd.resample`` does the rate change -- the same algorithm
restated (parity with the package unpinned: it is not in the build image), only reached when a file's rate
differs from the internal rate.
"""


import numpy as np

from .audio_io import pcm_channels, pcm_to_float, unpack24
from .config import Config
from .log import Code, ModuleError, debug, info, warning
from .utils import time_str


def _full_scale(array):
    """What 1.0 is in the array's own units: 2**(bits-1) for integer PCM (audio_io), 1 for floats."""
    return {np.dtype(np.int16): 32768.0, np.dtype(np.int32): 2147483648.0}.get(array.dtype, 1.0)


def count_max_peaks(array: np.ndarray):
    """dsp.py:49-54: the peak magnitude and how many samples sit on it (numpy.isclose), in two comparison
    passes over the array as it is -- float or integer PCM -- instead of isclose's float64 temporaries."""
    scale = _full_scale(array)
    low, high = float(array.min()), float(array.max())
    peak = abs(max(-low, high))
    max_value = peak / scale
    tol = (1e-8 + 1e-5 * max_value) * scale                     # isclose: |a - b| <= atol + rtol * |b|
    if peak - tol <= 0.0:                                        # silence: both signs' windows overlap
        hits = np.isclose(array, peak) | np.isclose(array, -peak)
        return max_value, int(np.count_nonzero(hits))
    if array.dtype.kind == "i":
        upper, lower = int(np.ceil(peak - tol)), int(np.floor(-peak + tol))
    else:
        upper, lower = array.dtype.type(peak - tol), array.dtype.type(-peak + tol)
        # (the threshold rounded to the array's type may admit a value a hair outside: settle those exactly)
        if float(upper) < peak - tol:
            upper = np.nextafter(upper, array.dtype.type(np.inf))
        if float(lower) > -peak + tol:
            lower = np.nextafter(lower, array.dtype.type(-np.inf))
    return max_value, int(np.count_nonzero(array >= upper)) + int(np.count_nonzero(array <= lower))


def _resample(array, sample_rate, required):
    try:                                            # pragma: no cover - not installed in the build image
        from resampy import resample

        return resample(array, sample_rate, required, axis=0)
    except ImportError:
        from .resample import resample     # the same algorithm restated (resampy is not a dependency one can count on)

        return resample(array, sample_rate, required)


LATER = object()      # check(..., peaks=LATER): the caller will hand the peak statistics to peak_warnings itself


def peak_warnings(peaks, config: Config) -> None:
    """checker.py:118-130: a target with many samples on its peak is clipped (peak at full scale) or has been
    through a limiter already."""
    max_value, max_count = peaks
    if max_count > config.clipping_samples_threshold:
        if np.isclose(max_value, 1.0):
            warning(Code.WARNING_TARGET_IS_CLIPPING)
        elif max_count > config.limited_samples_threshold:
            warning(Code.WARNING_TARGET_LIMITER_IS_APPLIED)


def check(array: np.ndarray, sample_rate: int, config: Config, name: str, peaks=None):
    """checker.py:90-137: returns the validated ``(array (n, 2), internal_sample_rate)``.  ``array`` may be
    integer PCM as a file holds it (audio_io); ``peaks`` = ``count_max_peaks`` of the track when the caller
    has it already (``process`` takes it on the GPU, ``mgx_peak_count``) -- the samples are then not read --
    or ``LATER`` when it will call ``peak_warnings`` itself."""
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

    channels = pcm_channels(array)
    converting = channels != 2 or sample_rate != config.internal_sample_rate
    if converting and peaks is LATER:
        peaks = None                                                    # (taken below, on the converted track)
    if array.dtype == np.uint8 and (converting or (target and peaks is None)):
        array = unpack24(array)                                         # packed samples: only as they are, or not
    if channels == 1:
        info(Code.INFO_TARGET_IS_MONO if target else Code.INFO_REFERENCE_IS_MONO)
        array = np.repeat(array, repeats=2, axis=1)                     # dsp.py:45-46
    elif channels != 2:
        raise ModuleError(Code.ERROR_TARGET_NUM_OF_CHANNELS_IS_EXCEEDED if target
                          else Code.ERROR_REFERENCE_NUM_OF_CHANNELS_IS_EXCEEDED)

    if sample_rate != config.internal_sample_rate:
        debug(f"{name}: converting {sample_rate} Hz -> {config.internal_sample_rate} Hz")
        # float64 like the arrays soundfile hands the reference's resampler (checker.py:42), also for files
        # that arrive as float32 (FLOAT WAVE, 8-bit PCM)
        array = _resample(np.asarray(pcm_to_float(array, np.float64), dtype=np.float64), sample_rate,
                          config.internal_sample_rate)
        if target:
            warning(Code.WARNING_TARGET_IS_RESAMPLED)
        else:
            info(Code.INFO_REFERENCE_IS_RESAMPLED)
        sample_rate = config.internal_sample_rate

    if target and peaks is not LATER:
        peak_warnings(peaks if peaks is not None else count_max_peaks(array), config)
    return array, sample_rate


def check_equality(target: np.ndarray, reference: np.ndarray) -> None:
    """checker.py:140-142: numpy.allclose of the two tracks, evaluated a block at a time so that tracks
    that differ -- the normal case -- are told apart after the first block."""
    if target.shape != reference.shape:
        return
    step = 1 << 18
    for i in range(0, target.shape[0], step):
        a = pcm_to_float(target[i:i + step], np.float64)
        b = pcm_to_float(reference[i:i + step], np.float64)
        if not np.allclose(a, b):
            return
    raise ModuleError(Code.ERROR_TARGET_EQUALS_REFERENCE)
