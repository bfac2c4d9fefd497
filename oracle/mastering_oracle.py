"""CPU oracle for the mastering hot path -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

Only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of
``bench.py`` may import this module.  The shipped path (``matchering_amd``)
never does: it runs on the HIP library and fails loudly without it.

What this is
------------
A float64 numpy/scipy restatement of ``matchering.stages.main`` (reference
v2.0.6) written as plain functions over arrays.  Each function cites the
reference lines it follows (paths relative to ``/root/reference``).  The scipy
routines the reference calls (``fftconvolve``, ``interp1d``, ``filtfilt``,
``lfilter``, ``butter``, ``maximum_filter1d``) are called here as well: scipy is
the reference's own numerical substrate (``requirements.txt:2``) and is
installed in this image (1.15.3).  The one dependency that is NOT installed is
statsmodels (``requirements.txt:5``, pinned only as ``>=0.13.2``); its LOWESS
(``it=0``) is restated in ``lowess_it0`` from the published algorithm
(Cleveland 1979 as implemented in ``statsmodels/nonparametric/_smoothers_lowess.pyx``).

Pinning
-------
The reference ships no tests and no golden vectors, so
parity is pinned by running the reference itself: ``oracle/reference_runner.py``
imports ``/root/reference/matchering`` unmodified (I/O modules stubbed) and
``tests/golden/make_golden.py`` compares this restatement with it and with the
compiled statsmodels LOWESS of ``/opt/conda/bin/python3.9``, then freezes the
outputs under ``tests/golden``.  ``tests/test_oracle_golden.py`` re-checks the
restatement against those frozen reference outputs on every run.
"""

from __future__ import annotations

import math
from types import SimpleNamespace

import numpy as np
from scipy import interpolate, signal
from scipy.ndimage import maximum_filter1d


# --------------------------------------------------------------------------
# parameters (matchering/defaults.py:25-155)
# --------------------------------------------------------------------------
def params(
    internal_sample_rate=44100,
    max_piece_size=15,
    threshold=(2**15 - 61) / 2**15,
    min_value=1e-6,
    fft_size=4096,
    lin_log_oversampling=4,
    rms_correction_steps=4,
    lowess_frac=0.0375,
    lowess_it=0,
    lowess_delta=0.001,
    attack=1.0,
    hold=1.0,
    release=3000.0,
    attack_filter_coefficient=-2.0,
    hold_filter_order=1,
    hold_filter_coefficient=7.0,
    release_filter_order=1,
    release_filter_coefficient=800.0,
):
    """Plain-namespace equivalent of ``Config()`` + ``LimiterConfig()``.

    ``max_piece_size`` is given in seconds and stored in samples, exactly as
    the reference constructor does (defaults.py:109)."""
    lim = SimpleNamespace(
        attack=attack,
        hold=hold,
        release=release,
        attack_filter_coefficient=attack_filter_coefficient,
        hold_filter_order=hold_filter_order,
        hold_filter_coefficient=hold_filter_coefficient,
        release_filter_order=release_filter_order,
        release_filter_coefficient=release_filter_coefficient,
    )
    return SimpleNamespace(
        internal_sample_rate=internal_sample_rate,
        max_piece_size=max_piece_size * internal_sample_rate,
        threshold=threshold,
        min_value=min_value,
        fft_size=fft_size,
        lin_log_oversampling=lin_log_oversampling,
        rms_correction_steps=rms_correction_steps,
        lowess_frac=lowess_frac,
        lowess_it=lowess_it,
        lowess_delta=lowess_delta,
        limiter=lim,
    )


# --------------------------------------------------------------------------
# level analysis  (stage_helpers/match_levels.py, dsp.py)
# --------------------------------------------------------------------------
def peak_normalize(x, threshold, eps, always):
    """dsp.py:93-100 ``normalize``: divide by max(eps, peak/threshold) when the
    peak is below the threshold (or unconditionally when ``always``)."""
    c = 1.0
    peak = np.abs(x).max()
    if peak < threshold or always:
        c = max(eps, peak / threshold)
    return x / c, c


def mid_side(x):
    """dsp.py:57-64 ``lr_to_ms``: mid = (L+R)*0.5, side = mid - R (same op order)."""
    mid = (x[:, 0] + x[:, 1]) * 0.5
    side = mid - x[:, 1]
    return mid, side


def piece_geometry(n, max_piece_size):
    """match_levels.py:47-59: divisions = int(n/max)+1, piece = int(n/divisions)."""
    divisions = int(n / max_piece_size) + 1
    return divisions, int(n / divisions)


def piece_rms(v, piece, divisions):
    """dsp.py:71-86 + match_levels.py:93-103: RMS of each of the ``divisions``
    leading pieces (tail dropped) and the RMS of those RMS values."""
    rows = v[: piece * divisions].reshape(divisions, piece)
    r = np.sqrt(np.einsum("ij,ij->i", rows, rows) / piece)
    return rows, r, math.sqrt(float(r @ r) / r.shape[0])


def loud_pieces(r, avg):
    """match_levels.py:62-71: indices with rms >= average, and the RMS of those."""
    idx = np.where(r >= avg)[0]
    sel = r[idx]
    return idx, math.sqrt(float(sel @ sel) / sel.shape[0])


def analyze(x, cfg):
    """match_levels.py:134-161 ``analyze_levels``."""
    mid, side = mid_side(x)
    divisions, piece = piece_geometry(mid.shape[0], cfg.max_piece_size)
    mrows, r, avg = piece_rms(mid, piece, divisions)
    srows = side[: piece * divisions].reshape(divisions, piece)
    idx, match = loud_pieces(r, avg)
    return SimpleNamespace(
        mid=mid, side=side, divisions=divisions, piece=piece, rmses=r,
        average_rms=avg, loud_idx=idx, match_rms=match,
        mid_loud=mrows[idx], side_loud=srows[idx],
    )


# --------------------------------------------------------------------------
# matching EQ  (stage_helpers/match_frequencies.py)
# --------------------------------------------------------------------------
def average_spectrum(pieces, fft_size):
    """match_frequencies.py:30-42.  scipy's boxcar / hop=F / unpadded STFT with
    'spectrum' scaling equals abs(rfft(segment))/F over the floor(piece/F)
    whole segments of every row; the mean runs over rows and segments."""
    k, plen = pieces.shape
    q = plen // fft_size
    segs = pieces[:, : q * fft_size].reshape(k, q, fft_size)
    return (np.abs(np.fft.rfft(segs, axis=-1)) / fft_size).mean(axis=(0, 1))


def lowess(y, frac, delta, it=0):
    """LOWESS with ``it`` robustness iterations on the index grid
    x = linspace(0, 1, n), as reached from dsp.py:103-106.

    Restated from statsmodels' ``_smoothers_lowess.pyx`` (not vendored in the
    reference): k = int(frac*n + 1e-10) nearest neighbours, tricube weights
    times the robustness weights, local linear fit; points closer than
    ``delta`` to the last fitted point are skipped and filled by linear
    interpolation.  After each pass the robustness weights become
    bisquare(|residual| / (6 median|residual|)) (Cleveland 1979), and the fit
    is repeated ``it`` more times."""
    y = np.asarray(y, dtype=np.float64)
    n = y.shape[0]
    x = np.linspace(0, 1, n)
    k = min(max(int(frac * n + 1e-10), 2), n)
    robust = np.ones(n)
    fit = np.zeros(n)
    for _ in range(it + 1):
        fit = np.zeros(n)
        i, last, lo, hi = 0, -1, 0, k
        while True:
            # slide the k-neighbourhood [lo, hi) to the right while that brings it closer
            while hi < n and x[i] > (x[lo] + x[hi]) / 2.0:
                lo += 1
                hi += 1
            radius = max(x[i] - x[lo], x[hi - 1] - x[i])
            xs = x[lo:hi]
            w = np.abs(xs - x[i]) / radius
            w = (1.0 - w * w * w) ** 3 * robust[lo:hi]
            sw = w.sum()
            if sw <= 0.0 or np.count_nonzero(w) == 1:
                fit[i] = y[i]
            else:
                w = w / sw
                xbar = float(np.sum(w * xs))
                dev = float(np.sum(w * (xs - xbar) ** 2))
                p = w * (1.0 + (x[i] - xbar) * (xs - xbar) / dev)
                fit[i] = float(np.sum(p * y[lo:hi]))
            if last < i - 1:
                a = (x[last + 1 : i] - x[last]) / (x[i] - x[last])
                fit[last + 1 : i] = a * fit[i] + (1.0 - a) * fit[last]
            last = i
            cut = x[last] + delta
            kk = last
            for kk in range(last + 1, n):
                if x[kk] > cut:
                    break
                if x[kk] == x[last]:
                    fit[kk] = fit[last]
                    last = kk
            i = max(kk - 1, last + 1)
            if last >= n - 1:
                break
        # robustness weights for the next pass
        r = np.abs(y - fit)
        med = np.median(r)
        if med == 0.0:
            r = (r > 0).astype(np.float64)
        else:
            r = r / (6.0 * med)
        r = np.minimum(r, 1.0)
        robust = (1.0 - r * r) ** 2
    return fit


def lowess_it0(y, frac, delta):
    return lowess(y, frac, delta, 0)


def smooth_matching_curve(h, cfg):
    """match_frequencies.py:45-75: cubic (not-a-knot) resample lin->log grid,
    LOWESS in index space, cubic resample back, then bins 0 and 1 are pinned."""
    half = cfg.fft_size // 2
    nyq = cfg.internal_sample_rate * 0.5
    g_lin = nyq * np.linspace(0, 1, half + 1)
    g_log = nyq * np.logspace(np.log10(4 / cfg.fft_size), 0, half * cfg.lin_log_oversampling + 1)
    h_log = interpolate.interp1d(g_lin, h, "cubic")(g_log)
    h_log_s = lowess(h_log, cfg.lowess_frac, cfg.lowess_delta, cfg.lowess_it)
    out = interpolate.interp1d(g_log, h_log_s, "cubic", fill_value="extrapolate")(g_lin)
    out[0] = 0
    out[1] = h[1]
    return out


def design_fir(target_pieces, reference_pieces, cfg):
    """match_frequencies.py:78-101 ``get_fir``."""
    a_t = average_spectrum(target_pieces, cfg.fft_size)
    a_r = average_spectrum(reference_pieces, cfg.fft_size)
    h = a_r / np.maximum(cfg.min_value, a_t)
    h_s = smooth_matching_curve(h, cfg)
    taps = np.fft.irfft(h_s)
    taps = np.fft.ifftshift(taps) * signal.windows.hann(taps.shape[0])
    return taps, SimpleNamespace(avg_target=a_t, avg_reference=a_r, raw=h, smooth=h_s)


def convolve_same(mid, mid_fir, side, side_fir):
    """match_frequencies.py:104-119: 'same'-mode FFT convolution of mid and
    side, then dsp.py:67-68 ``ms_to_lr``.  Returns (result (N,2), result_mid)."""
    ym = signal.fftconvolve(mid, mid_fir, "same")
    ys = signal.fftconvolve(side, side_fir, "same")
    # the reference's memory layout too (np.vstack(...).T: planar, i.e. an F-ordered (N,2) view) -- it is
    # what the later per-channel reductions of the reference run on, and so what their CPU time depends on
    return np.vstack((ym + ys, ym - ys)).T, ym


# --------------------------------------------------------------------------
# Hyrax limiter  (limiter/hyrax.py)
# --------------------------------------------------------------------------
def limiter_envelopes(y, cfg):
    """hyrax.py:78-99 up to the gain envelope.  Returns None when the limiter
    early-outs (hyrax.py:83-85), else a namespace with every intermediate."""
    thr = cfg.threshold
    sr = cfg.internal_sample_rate
    lim = cfg.limiter
    rect = np.abs(y).max(1)                      # dsp.py:117-121
    rect[rect <= thr] = thr
    rect /= thr
    if np.all(np.isclose(rect, 1.0)):
        return None
    g0 = 1.0 - 1.0 / rect                        # dsp.py:113-114 on 1/rectified

    attack = int(sr * lim.attack * 1e-3)         # utils.py:50-51
    w = attack if attack & 1 else attack + 1     # utils.py:54-55
    slided = maximum_filter1d(g0, size=2 * w - 1)            # hyrax.py:35-37
    rho = math.exp(lim.attack_filter_coefficient / attack)   # hyrax.py:48
    g_att = signal.filtfilt([1 - rho], [1, -rho], slided)    # hyrax.py:51

    hold = int(sr * lim.hold * 1e-3)
    half = (hold - 1) // 2                                   # hyrax.py:38-40
    sh = maximum_filter1d(np.pad(slided, (half, 0)), size=hold)[:-half]
    b1, a1 = signal.butter(lim.hold_filter_order, lim.hold_filter_coefficient, fs=sr)
    ho = signal.lfilter(b1, a1, sh)                          # hyrax.py:61-66
    b2, a2 = signal.butter(lim.release_filter_order,
                           lim.release_filter_coefficient / lim.release, fs=sr)
    ro = signal.lfilter(b2, a2, np.maximum(sh, ho))          # hyrax.py:68-73
    g_rel = np.maximum(ho, ro)                               # hyrax.py:75
    gain = 1.0 - np.maximum(np.maximum(g0, g_att), g_rel)    # hyrax.py:97 (dsp.py:124-125 max_mix; max is exact in any order)
    return SimpleNamespace(g0=g0, slided=slided, g_att=g_att, held=sh, hold_out=ho,
                           release_out=ro, g_rel=g_rel, gain=gain)


def limit(y, cfg):
    """hyrax.py:78-99 ``limit``."""
    env = limiter_envelopes(y, cfg)
    if env is None:
        return y
    return y * env.gain[:, None]


# --------------------------------------------------------------------------
# the pipeline  (stages.py:210-272)
# --------------------------------------------------------------------------
def master(target, reference, cfg, need_default=True, need_no_limiter=False,
           need_no_limiter_normalized=False, trace=None, fir=None):
    """Float64 restatement of ``matchering.stages.main``.  ``trace`` (a dict)
    receives the intermediates the per-stage parity tests compare against.
    ``fir`` = (mid taps, side taps) replaces the FIR pair of stage 2 (not a reference
    feature: the checker of the album mode of matchering_amd/batch.py)."""
    target = np.asarray(target, dtype=np.float64)
    reference = np.asarray(reference, dtype=np.float64)
    eps = cfg.min_value

    # stage 1, stages.py:38-104
    reference, final_c = peak_normalize(reference, cfg.threshold, eps, always=False)
    t = analyze(target, cfg)
    r = analyze(reference, cfg)
    c0 = r.match_rms / max(eps, t.match_rms)
    t_mid, t_side = t.mid * c0, t.side * c0

    # stage 2, stages.py:107-135
    fir_mid, dm = design_fir(t.mid_loud * c0, r.mid_loud, cfg)
    fir_side, ds = design_fir(t.side_loud * c0, r.side_loud, cfg)
    if fir is not None:
        fir_mid, fir_side = np.asarray(fir[0], dtype=np.float64), np.asarray(fir[1], dtype=np.float64)
    y, y_mid = convolve_same(t_mid, fir_mid, t_side, fir_side)

    # stage 3, stages.py:138-170
    coeffs = []
    for _ in range(cfg.rms_correction_steps):
        clipped = np.clip(y_mid, -1.0, 1.0)
        _, rm, avg = piece_rms(clipped, t.piece, t.divisions)
        _, m = loud_pieces(rm, avg)
        c = r.match_rms / max(eps, m)
        coeffs.append(c)
        y_mid = y_mid * c
        y = y * c

    # stage 4, stages.py:173-207
    out_norm = None
    norm_c = None
    if need_no_limiter_normalized:
        out_norm, norm_c = peak_normalize(y, cfg.threshold, eps, always=True)
    out = None
    if need_default:
        out = limit(y, cfg) * final_c
    if trace is not None:
        trace.update(
            final_amplitude_coefficient=final_c, rms_coefficient=c0,
            target_divisions=t.divisions, target_piece=t.piece, target_rmses=t.rmses,
            target_match_rms=t.match_rms, target_loud_idx=t.loud_idx,
            reference_divisions=r.divisions, reference_piece=r.piece,
            reference_rmses=r.rmses, reference_match_rms=r.match_rms,
            reference_loud_idx=r.loud_idx,
            mid=dm, side=ds, fir_mid=fir_mid, fir_side=fir_side,
            correction_coefficients=np.array(coeffs), normalize_coefficient=norm_c,
            result_no_limiter=y,
        )
    return out, (y if need_no_limiter else None), out_norm
