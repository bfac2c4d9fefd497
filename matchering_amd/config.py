"""``Config`` / ``LimiterConfig``: the knobs of the mastering pipeline.

Same constructor keywords, attribute names, unit conventions and validation as
matchering/defaults.py:25-155 (a user's ``mg.Config(...)`` call keeps working):
times given in seconds are stored in samples (``max_piece_size``, ``preview_*``),
bad values raise ``AssertionError``.  ``to_native()`` marshals the fields the GPU
path consumes into the ``mgx_config`` struct of include/mgx.h.
"""

import math

from .log import debug

_LIMITER_FIELDS = ("attack", "hold", "release", "attack_filter_coefficient", "hold_filter_order",
                   "hold_filter_coefficient", "release_filter_order", "release_filter_coefficient")


def _require(condition, what):
    assert condition, what


class LimiterConfig:
    """Timing (milliseconds) and filter design of the Hyrax limiter (defaults.py:25-58)."""

    def __init__(self, attack: float = 1, hold: float = 1, release: float = 3000,
                 attack_filter_coefficient: float = -2, hold_filter_order: int = 1,
                 hold_filter_coefficient: float = 7, release_filter_order: int = 1,
                 release_filter_coefficient: float = 800):
        for label, value in (("attack", attack), ("hold", hold), ("release", release)):
            _require(value > 0, f"limiter {label} must be positive")
        for label, value in (("hold_filter_order", hold_filter_order),
                             ("release_filter_order", release_filter_order)):
            _require(isinstance(value, int) and value > 0, f"{label} must be a positive int")
        self.attack, self.hold, self.release = attack, hold, release
        self.attack_filter_coefficient = attack_filter_coefficient
        self.hold_filter_order = hold_filter_order
        self.hold_filter_coefficient = hold_filter_coefficient
        self.release_filter_order = release_filter_order
        self.release_filter_coefficient = release_filter_coefficient

    def __repr__(self):
        return "LimiterConfig(" + ", ".join(f"{k}={getattr(self, k)!r}" for k in _LIMITER_FIELDS) + ")"


class Config:
    """Pipeline parameters (defaults.py:61-155)."""

    def __init__(self, internal_sample_rate: int = 44100, max_length: float = 15 * 60,
                 max_piece_size: float = 15, threshold: float = (2 ** 15 - 61) / 2 ** 15,
                 min_value: float = 1e-6, fft_size: int = 4096, lin_log_oversampling: int = 4,
                 rms_correction_steps: int = 4, clipping_samples_threshold: int = 8,
                 limited_samples_threshold: int = 128, allow_equality: bool = False,
                 lowess_frac: float = 0.0375, lowess_it: int = 0, lowess_delta: float = 0.001,
                 preview_size: float = 30, preview_analysis_step: float = 5,
                 preview_fade_size: float = 1, preview_fade_coefficient: float = 8,
                 temp_folder: str = None, limiter: LimiterConfig = None):
        sr = internal_sample_rate
        _require(isinstance(sr, int) and sr > 0, "internal_sample_rate must be a positive int")
        if sr != 44100:
            debug(f"internal_sample_rate {sr}: the reference only vouches for 44100; this path is tested from 8 kHz to 192 kHz")
        seconds_per_fft = fft_size / sr
        _require(max_length > 0 and max_length > seconds_per_fft, "max_length too small")
        _require(min_value < threshold < 1, "threshold must lie in (min_value, 1)")
        _require(0 < min_value < 0.1, "min_value must lie in (0, 0.1)")
        _require(max_piece_size > 0 and seconds_per_fft < max_piece_size < max_length,
                 "max_piece_size must lie in (fft_size / sample_rate, max_length)")
        _require(fft_size > 1 and math.log2(fft_size).is_integer(), "fft_size must be a power of two")
        _require(isinstance(lin_log_oversampling, int) and lin_log_oversampling > 0,
                 "lin_log_oversampling must be a positive int")
        _require(isinstance(rms_correction_steps, int) and rms_correction_steps >= 0,
                 "rms_correction_steps must be a non-negative int")
        _require(isinstance(clipping_samples_threshold, int) and isinstance(limited_samples_threshold, int),
                 "sample-count thresholds must be ints")
        _require(0 <= clipping_samples_threshold < limited_samples_threshold and limited_samples_threshold > 0,
                 "need 0 <= clipping_samples_threshold < limited_samples_threshold")
        _require(isinstance(allow_equality, bool), "allow_equality must be a bool")
        _require(lowess_frac > 0 and lowess_delta >= 0 and isinstance(lowess_it, int) and lowess_it >= 0,
                 "bad LOWESS parameters")
        _require(preview_size > 5 and preview_analysis_step > 1 and preview_fade_size > 0
                 and preview_fade_coefficient >= 2, "bad preview parameters")
        _require(temp_folder is None or isinstance(temp_folder, str), "temp_folder must be a str")
        limiter = LimiterConfig() if limiter is None else limiter
        _require(isinstance(limiter, LimiterConfig), "limiter must be a LimiterConfig")

        self.internal_sample_rate = sr
        self.max_length = max_length
        self.threshold = threshold
        self.min_value = min_value
        self.max_piece_size = max_piece_size * sr            # samples from here on
        self.fft_size = fft_size
        self.lin_log_oversampling = lin_log_oversampling
        self.rms_correction_steps = rms_correction_steps
        self.clipping_samples_threshold = clipping_samples_threshold
        self.limited_samples_threshold = limited_samples_threshold
        self.allow_equality = allow_equality
        self.lowess_frac, self.lowess_it, self.lowess_delta = lowess_frac, lowess_it, lowess_delta
        self.preview_size = preview_size * sr
        self.preview_analysis_step = preview_analysis_step * sr
        self.preview_fade_size = preview_fade_size * sr
        self.preview_fade_coefficient = preview_fade_coefficient
        self.temp_folder = temp_folder
        self.limiter = limiter

    def to_native(self):
        """The ``mgx_config`` struct for this configuration."""
        from ._native import MgxConfig

        lim = self.limiter
        return MgxConfig(
            internal_sample_rate=self.internal_sample_rate, fft_size=self.fft_size,
            lin_log_oversampling=self.lin_log_oversampling, rms_correction_steps=self.rms_correction_steps,
            max_piece_size=float(self.max_piece_size), threshold=float(self.threshold),
            min_value=float(self.min_value), lowess_frac=float(self.lowess_frac), lowess_it=self.lowess_it,
            lowess_delta=float(self.lowess_delta), attack_ms=float(lim.attack), hold_ms=float(lim.hold),
            release_ms=float(lim.release), attack_filter_coefficient=float(lim.attack_filter_coefficient),
            hold_filter_order=lim.hold_filter_order, release_filter_order=lim.release_filter_order,
            hold_filter_coefficient=float(lim.hold_filter_coefficient),
            release_filter_coefficient=float(lim.release_filter_coefficient))
