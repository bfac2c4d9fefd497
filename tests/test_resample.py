"""Sample-rate conversion of off-rate files (checker.py:30-45): matchering_amd/resample.py against the literal
restatement of resampy's loops (oracle/resampy_oracle.py) and against what band-limited interpolation must do.
resampy itself is not in this image: parity with the package is unpinned, and both files say so."""
import numpy as np
import pytest

import resampy_oracle
from matchering_amd import resample as product


def test_the_filter_is_resampys_kaiser_best():
    win, num_table = product.kaiser_best()
    want, bits, rolloff = resampy_oracle.sinc_window()
    assert num_table == bits == 512 and win.shape == (64 * 512 + 1,)
    assert np.abs(win - want).max() <= 1e-15            # np.i0 and scipy's kaiser are the same window
    assert abs(win[0] - rolloff) <= 1e-15 and abs(win[-1]) <= 1e-7
    # (the sinc's zero crossings lie at multiples of 1 / rolloff, not at the table's 512-entry marks)
    assert abs(win[int(round(512 * 10 / rolloff))]) <= 1e-4


@pytest.mark.parametrize("sr,new,n", [(48000, 44100, 700), (48000, 44100, 3000), (44100, 48000, 3000), (96000, 44100, 6000),
                                      (22050, 44100, 2000), (44100, 44101, 300), (8000, 44100, 900), (192000, 44100, 9000),
                                      (44100, 22050, 1000), (48000, 44100, 5)])
def test_vectorised_form_equals_the_loops(sr, new, n):
    """The product evaluates resampy's sum as a polyphase filter (one prototype holding the weights of every phase of
    the rational ratio, scipy's upfirdn), or literally, sample by sample, when the ratio has too many phases; both
    must be the loops of the restatement, ends of the array included."""
    rng = np.random.RandomState(sr % 1000 + n)
    x = rng.randn(n, 2)
    got = product.resample(x, sr, new, block=257)
    want = resampy_oracle.resample(x, sr, new)
    assert got.shape == want.shape == (int(n * (float(new) / sr)), 2)
    assert np.abs(got - want).max() <= 1e-11
    mono = product.resample(x[:, 0], sr, new)
    assert mono.shape == (want.shape[0],) and np.abs(mono - want[:, 0]).max() <= 1e-11
    assert np.abs(product.resample(x, sr, new, max_phases=0) - want).max() <= 1e-11      # all literal


def test_exact_phases_against_resampys_rounded_times_far_into_a_file():
    """resampy's t * (1 / ratio) is rounded; the phase arithmetic of the fast path is exact.  Twenty seconds in, the
    two differ by what that rounding moves the table position: far below a float32 sample's resolution."""
    x = np.random.RandomState(9).randn(48000 * 21, 2)
    fast = product.resample(x, 48000, 44100)
    literal = np.zeros_like(fast)
    t = np.arange(44100 * 20, 44100 * 20 + 3000)
    product._literal(product._Plan(48000, 44100), x, t, literal)
    assert 0.0 < np.abs(fast[t] - literal[t]).max() <= 1e-8


@pytest.mark.parametrize("sr,new", [(48000, 44100), (44100, 48000), (96000, 44100), (32000, 44100)])
def test_a_sine_stays_the_same_sine(sr, new):
    """1 kHz and 9 kHz at the old rate are 1 kHz and 9 kHz at the new one, to the filter's pass-band ripple, away
    from the ends of the array (where half of the filter has nothing to read)."""
    n = sr // 4
    t_old = np.arange(n) / sr
    x = np.stack([np.sin(2 * np.pi * 1000 * t_old), 0.5 * np.cos(2 * np.pi * 9000 * t_old)], axis=1)
    y = product.resample(x, sr, new)
    t_new = np.arange(y.shape[0]) / new
    want = np.stack([np.sin(2 * np.pi * 1000 * t_new), 0.5 * np.cos(2 * np.pi * 9000 * t_new)], axis=1)
    edge = 200
    assert np.abs(y[edge:-edge] - want[edge:-edge]).max() <= 5e-4      # (the table's pass-band gain is 1 + 1.7e-4)


def test_the_checker_resamples_with_it(monkeypatch):
    import sys

    from matchering_amd import checker

    monkeypatch.setitem(sys.modules, "resampy", None)    # (as in this image: not importable)
    x = np.random.RandomState(3).randn(1000, 2)
    got = checker._resample(x, 48000, 44100)
    assert got.shape == (918, 2) and np.abs(got - product.resample(x, 48000, 44100)).max() == 0.0
