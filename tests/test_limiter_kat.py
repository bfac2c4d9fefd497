"""The reference's own limiter stimulus (LIMITER_TEST.md:3-4): a 440 Hz sine whose envelope exceeds 0 dB
in some regions.  The document shows pictures, not numbers, so the known answers are the properties it
illustrates: a brick wall at the threshold, untouched material where the envelope stays below it
(after the release has decayed), and no flat tops -- the limited regions keep the sine's shape because
the gain moves slowly (attack/hold/release, hyrax.py:43-99) instead of clipping.
"""

import numpy as np
import pytest

import mastering_oracle as mo
from conftest import rms_error

RATE = 44100


def stimulus():
    t = np.arange(int(6.0 * RATE)) / RATE
    envelope = 0.6 + 0.9 * np.exp(-((t - 1.5) / 0.25) ** 2) + 0.7 * (np.abs(t - 4.0) < 0.4)
    tone = envelope * np.sin(2 * np.pi * 440.0 * t)
    return t, envelope, np.stack([tone, 0.9 * tone], axis=1)


def check_properties(t, envelope, x, y, threshold):
    assert np.abs(y).max() <= threshold * (1 + 1e-5)
    assert np.abs(x).max() > 1.3                                   # the stimulus does exceed 0 dB
    quiet = (t < 0.4)                                              # before the first loud region
    assert np.abs(y[quiet] - x[quiet]).max() <= 1e-5
    # gain actually applied in the loud regions: slow and smooth, never a per-sample clamp
    loud = np.abs(x[:, 0]) > 0.5
    gain = np.ones(len(t))
    gain[loud] = y[loud, 0] / x[loud, 0]
    held = (np.abs(t - 4.0) < 0.3) & loud
    assert gain[held].max() - gain[held].min() <= 0.02             # steady inside the plateau: hold + release
    assert np.abs(np.diff(gain[np.flatnonzero(held)])).max() <= 1e-3
    # shape kept: the limited plateau is still a sine of ONE amplitude, not a flat-topped one
    seg = y[held, 0]
    k = np.arange(len(seg))
    basis = np.stack([np.sin(2 * np.pi * 440 * t[held]), np.cos(2 * np.pi * 440 * t[held])], axis=1)
    coef, *_ = np.linalg.lstsq(basis, seg, rcond=None)
    residual = seg - basis @ coef
    assert np.sqrt(np.mean(residual ** 2)) <= 0.01 * np.sqrt(np.mean(seg ** 2))     # < 1 % distortion
    assert k.size > 1000


def test_oracle_limiter_known_answers():
    t, envelope, x = stimulus()
    cfg = mo.params()
    check_properties(t, envelope, x, mo.limit(x, cfg), cfg.threshold)


@pytest.mark.gpu
def test_gpu_limiter_known_answers():
    import matchering_amd as mg
    from matchering_amd import kernels

    t, envelope, x = stimulus()
    x32 = x.astype(np.float32)
    out, active = kernels.limit(x32, mg.Config(), gain=1.0, post_gain=1.0)
    assert active
    want = mo.limit(x32.astype(np.float64), mo.params())
    assert rms_error(out, want) <= 1e-6
    check_properties(t, envelope, x32.astype(np.float64), out.astype(np.float64), mg.Config().threshold)
