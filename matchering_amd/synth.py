"""Deterministic synthetic stereo programme material.

There are no audio files in this environment, so tests, the golden fixtures and
``bench.py`` all draw their inputs from here.  The signal is music-shaped on
purpose: a low-passed noise bed (6th-order roll-off above 2 kHz), a 220 Hz tone
with different left/right weights (so the side channel is not empty), a -70 dB
white floor (so every FFT bin carries signal, as recorded material does) and a
7-second loudness cycle, so that piece RMS values straddle their average with
clear margins and the loud-piece selection is not a coin flip.

Values are produced in float64 and rounded ONCE to float32; callers hand the
same float32 values to the GPU path and (promoted to float64) to the oracle.
"""

import numpy as np
from scipy import signal


def synth(seconds: float, sample_rate: int = 44100, seed: int = 1,
          corner: float = 2000.0) -> np.ndarray:
    """(n, 2) float64 programme material, n = int(seconds * sample_rate)."""
    n = int(seconds * sample_rate)
    rng = np.random.RandomState(seed)
    x = 0.35 * rng.randn(n, 2)
    b, a = signal.butter(2, min(corner, 0.2 * sample_rate), fs=sample_rate)
    xt = np.ascontiguousarray(x.T)                   # (one channel contiguous at a time: same arithmetic, less striding)
    for _ in range(3):
        xt = signal.lfilter(b, a, xt, axis=1)
    x = np.ascontiguousarray(xt.T)
    t = np.arange(n) / sample_rate
    x += 0.2 * np.sin(2 * np.pi * 220.0 * t)[:, None] * np.array([1.0, 0.7])
    x += 3e-4 * rng.randn(n, 2)
    x *= (0.5 + 0.5 * np.sin(2 * np.pi * t / 7.0) ** 2)[:, None]
    return x


def make_pair(seconds: float, sample_rate: int = 44100, pair: int = 0,
              reference_seconds: float = None, reference_gain: float = 2.5,
              target_gain: float = 0.5, reference_corner: float = 3500.0):
    """Target/reference pair number ``pair`` as float32 (n, 2) arrays.

    target    = target_gain * synth(seed 1+2*pair)            (quiet, unclipped)
    reference = clip(reference_gain * synth(seed 2+2*pair))   (loud, clipped)
    ``reference_gain`` 2.5 limits a fraction of a percent of the frames of the
    result, 6.0 ("hot") several percent; values below ~0.8 leave the reference
    under the threshold and exercise the final-amplitude branch instead.  The
    reference's noise bed is brighter (``reference_corner`` Hz against the
    target's 2 kHz) so the matching EQ has real work to do."""
    rs = seconds if reference_seconds is None else reference_seconds
    target = target_gain * synth(seconds, sample_rate, seed=1 + 2 * pair)
    reference = np.clip(reference_gain * synth(rs, sample_rate, seed=2 + 2 * pair, corner=reference_corner), -1.0, 1.0)
    return target.astype(np.float32), reference.astype(np.float32)
