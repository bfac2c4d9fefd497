"""TEST INFRASTRUCTURE -- a restatement of resampy's band-limited sinc interpolation, the resampler the reference
calls for files that are not at ``internal_sample_rate`` (matchering/checker.py:22,42:
``resampy.resample(array, sample_rate, required_sample_rate, axis=0)``, default filter ``kaiser_best``).

PARITY UNPINNED.  resampy is a third-party dependency (requirements.txt:4, ``resampy>=0.4.2``) that is absent
from /root/reference and from every interpreter of this image, and the reference's tests hold no vector for
it, so nothing here could be checked against the package itself.  What is restated is the published algorithm
of resampy 0.4.2 (J. O. Smith's "Digital Audio Resampling", the windowed-sinc table with linear interpolation
between table entries):

  * the filter ``kaiser_best``: ``resampy.filters.sinc_window(num_zeros=64, precision=9,
    window=kaiser(beta=14.769656459379492), rolloff=0.9475937167399596)`` -- the right half of a Kaiser-windowed
    sinc with 64 zero crossings at 2**9 = 512 table entries per crossing (the package ships this table as
    ``data/kaiser_best.npz``; its documentation gives the parameters);
  * ``resampy.core.resample``: n_out = int(n * ratio); the table is scaled by the ratio when down-sampling;
    output sample t sits at input time t / ratio;
  * ``resampy.interpn.resample_f``: left and right wing of the filter, table stride int(scale * 512), weights
    interpolated linearly between neighbouring table entries (``interp_delta``).

Only tests may import this file (plain Python loops: small cases only).  The product's vectorised form is
matchering_amd/resample.py; tests/test_resample.py holds the two against each other and both against what a
band-limited resampler must do to a sine.
"""
import numpy as np
from scipy.signal.windows import kaiser

NUM_ZEROS = 64
PRECISION = 9
BETA = 14.769656459379492
ROLLOFF = 0.9475937167399596


def sinc_window(num_zeros=NUM_ZEROS, precision=PRECISION, beta=BETA, rolloff=ROLLOFF):
    """resampy.filters.sinc_window: (half window, table entries per zero crossing, roll-off)."""
    num_bits = 2 ** precision
    n = num_bits * num_zeros
    sinc_win = rolloff * np.sinc(rolloff * np.linspace(0, num_zeros, num=n + 1, endpoint=True))
    taper = kaiser(2 * n + 1, beta)[n:]
    return taper * sinc_win, num_bits, rolloff


def resample(x, sr_orig, sr_new):
    """resampy.core.resample along axis 0 with filter='kaiser_best', as plain loops (float64)."""
    x = np.asarray(x, dtype=np.float64)
    ratio = float(sr_new) / sr_orig
    n_out = int(x.shape[0] * ratio)
    interp_win, num_table, _ = sinc_window()
    if ratio < 1:
        interp_win = interp_win * ratio
    interp_delta = np.zeros_like(interp_win)
    interp_delta[:-1] = np.diff(interp_win)
    scale = min(1.0, ratio)
    time_increment = 1.0 / ratio
    index_step = int(scale * num_table)
    nwin = interp_win.shape[0]
    n_orig = x.shape[0]
    y = np.zeros((n_out,) + x.shape[1:], dtype=np.float64)
    for t in range(n_out):
        time_register = t * time_increment
        n = int(time_register)
        frac = scale * (time_register - n)
        index_frac = frac * num_table
        offset = int(index_frac)
        eta = index_frac - offset
        i_max = min(n + 1, (nwin - offset) // index_step)
        for i in range(i_max):
            weight = interp_win[offset + i * index_step] + eta * interp_delta[offset + i * index_step]
            y[t] += weight * x[n - i]
        frac = scale - frac
        index_frac = frac * num_table
        offset = int(index_frac)
        eta = index_frac - offset
        k_max = min(n_orig - n - 1, (nwin - offset) // index_step)
        for k in range(k_max):
            weight = interp_win[offset + k * index_step] + eta * interp_delta[offset + k * index_step]
            y[t] += weight * x[n + k + 1]
    return y
