"""Kernel index arithmetic on the CPU: tests/emu/libmgx_emu.so drives the SAME per-thread phase
functions the HIP kernels inline (matchering_amd/csrc/*_kernel.h), with a loop over thread ids
where the GPU has a workgroup.  Checked against the oracle here so that a GPU run is spent on
performance, not on off-by-one hunting.  The emulation is test infrastructure only: the shipped
library (libmgx.so) contains no host implementation of any kernel.
"""

import ctypes
import importlib.util
import os

import numpy as np
import pytest

import mastering_oracle as mo
from cases import CASES, build_inputs, oracle_params
from conftest import ROOT, rms_error

c_float_p = ctypes.POINTER(ctypes.c_float)
c_double_p = ctypes.POINTER(ctypes.c_double)


@pytest.fixture(scope="module")
def emu():
    spec = importlib.util.spec_from_file_location("mgx_emu_build", os.path.join(ROOT, "tests", "emu", "build.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return ctypes.CDLL(mod.build())


def _fp(a):
    return a.ctypes.data_as(c_float_p)


def _dp(a):
    return a.ctypes.data_as(c_double_p)


def emu_convolve(emu, x, hm, hs, gain=1.0):
    n = x.shape[0]
    x = np.ascontiguousarray(x, dtype=np.float32)
    y = np.zeros((n, 2), dtype=np.float32)
    ymid = np.zeros(n, dtype=np.float32)
    peak = ctypes.c_double()
    hm = np.ascontiguousarray(hm, dtype=np.float64)
    hs = np.ascontiguousarray(hs, dtype=np.float64)
    rc = emu.emu_convolve(_fp(x), ctypes.c_longlong(n), _dp(hm), _dp(hs), ctypes.c_int(len(hm)),
                          ctypes.c_double(gain), _fp(y), _fp(ymid), ctypes.byref(peak))
    assert rc == 0
    return y, ymid, peak.value


@pytest.mark.parametrize("taps,n", [(64, 1), (64, 333), (128, 1000), (256, 2100), (512, 2048 + 77),
                                    (1024, 5000), (2048, 9001), (4096, 3 * 8192 + 5)])
def test_convolution_phases(emu, taps, n):
    rng = np.random.RandomState(taps + n)
    x = (0.3 * rng.randn(n, 2)).astype(np.float32)
    hm, hs = rng.randn(taps) / np.sqrt(taps), rng.randn(taps) / np.sqrt(taps)
    y, ymid, peak = emu_convolve(emu, x, hm, hs, gain=1.7)
    mid, side = mo.mid_side(x.astype(np.float64))
    want, want_mid = mo.convolve_same(mid * 1.7, hm, side * 1.7, hs)
    assert rms_error(y, want) <= 1e-6
    assert rms_error(ymid, want_mid) <= 1e-6
    assert abs(peak - np.abs(y).max()) <= 1e-6


@pytest.mark.parametrize("taps,block_log2,n", [(256, 8, 1500), (1024, 9, 4000), (2048, 10, 7001),
                                               (512, 7, 1), (4096, 11, 9000)])
def test_partitioned_convolution_phases(emu, taps, block_log2, n):
    """Filters longer than half a block: uniformly partitioned overlap-save (K = 2*taps/N partitions),
    the path a 16 k-tap FIR takes on N = 8192 blocks; checked here at small N against the same
    direct convolution."""
    rng = np.random.RandomState(taps + n)
    x = np.ascontiguousarray((0.3 * rng.randn(n, 2)).astype(np.float32))
    hm, hs = rng.randn(taps) / np.sqrt(taps), rng.randn(taps) / np.sqrt(taps)
    y = np.zeros((n, 2), dtype=np.float32)
    ymid = np.zeros(n, dtype=np.float32)
    peak = ctypes.c_double()
    rc = emu.emu_convolve_blocked(_fp(x), ctypes.c_longlong(n), _dp(hm), _dp(hs), ctypes.c_int(taps),
                                  ctypes.c_double(0.8), _fp(y), _fp(ymid), ctypes.byref(peak),
                                  ctypes.c_int(block_log2))
    assert rc == 0
    mid, side = mo.mid_side(x.astype(np.float64))
    want, want_mid = mo.convolve_same(mid * 0.8, hm, side * 0.8, hs)
    assert rms_error(y, want) <= 1e-6
    assert rms_error(ymid, want_mid) <= 1e-6
    assert abs(peak.value - np.abs(y).max()) <= 1e-6


@pytest.mark.parametrize("taps,n,run", [(2048, 1, 1), (2048, 1024, 1), (2048, 9001, 3), (4096, 30000, 4),
                                        (8192, 8 * 4096 + 77, 2), (16384, 5 * 8192 + 1234, 3), (16384, 3000, 11)])
def test_delay_line_convolution_phases(emu, taps, n, run):
    """A filter of two partitions (taps = N, config #5's 16384 taps) as a frequency-domain delay line:
    mid + j side in ONE transform per block, un-mixed through the mirror bin at the multiply, partition 1's
    product carried to the next block of the workgroup's run (conv_delay_kernel.h).  Against the same direct
    convolution as the kernels above, over runs that end inside and beyond the track."""
    rng = np.random.RandomState(taps + n)
    x = np.ascontiguousarray((0.3 * rng.randn(n, 2)).astype(np.float32))
    hm, hs = rng.randn(taps) / np.sqrt(taps), rng.randn(taps) / np.sqrt(taps)
    y = np.zeros((n, 2), dtype=np.float32)
    ymid = np.zeros(n, dtype=np.float32)
    nblocks = (n + taps // 2 - 1) // (taps // 2)
    peaks = np.zeros(nblocks, dtype=np.float32)
    rc = emu.emu_convolve_delay(_fp(x), ctypes.c_longlong(n), _dp(hm), _dp(hs), ctypes.c_int(taps),
                                ctypes.c_double(0.8), _fp(y), _fp(ymid), _fp(peaks), ctypes.c_int(run))
    assert rc == 0
    mid, side = mo.mid_side(x.astype(np.float64))
    want, want_mid = mo.convolve_same(mid * 0.8, hm, side * 0.8, hs)
    assert rms_error(y, want) <= 1e-6
    assert rms_error(ymid, want_mid) <= 1e-6
    hop = taps // 2
    for b in range(nblocks):
        assert abs(peaks[b] - np.abs(y[b * hop:(b + 1) * hop]).max()) <= 1e-6


@pytest.mark.parametrize("taps,n", [(512, 1), (512, 1535), (512, 1536), (1024, 9001), (2048, 30000),
                                    (4096, 12288), (4096, 5 * 12288 + 4321)])
def test_wide_block_convolution_phases(emu, taps, n):
    """F taps on N = 4F blocks (conv_wide_kernel.h: three quarters of a block are fresh output; mid + j side in one
    transform, pairs of bins un-mixed through the mirror bin): the phase functions of k_conv_wide on the 2048-
    to 16384-point plans, against the same direct convolution as the other kernels; block peaks too."""
    rng = np.random.RandomState(taps + n)
    x = np.ascontiguousarray((0.3 * rng.randn(n, 2)).astype(np.float32))
    hm, hs = rng.randn(taps) / np.sqrt(taps), rng.randn(taps) / np.sqrt(taps)
    y = np.zeros((n, 2), dtype=np.float32)
    ymid = np.zeros(n, dtype=np.float32)
    hop = 3 * taps
    nblocks = (n + hop - 1) // hop
    peaks = np.zeros(nblocks, dtype=np.float32)
    rc = emu.emu_convolve_wide(_fp(x), ctypes.c_longlong(n), _dp(hm), _dp(hs), ctypes.c_int(taps),
                               ctypes.c_double(1.3), _fp(y), _fp(ymid), _fp(peaks))
    assert rc == 0
    mid, side = mo.mid_side(x.astype(np.float64))
    want, want_mid = mo.convolve_same(mid * 1.3, hm, side * 1.3, hs)
    assert rms_error(y, want) <= 1e-6
    assert rms_error(ymid, want_mid) <= 1e-6
    for b in range(nblocks):
        assert abs(peaks[b] - np.abs(y[b * hop:(b + 1) * hop]).max()) <= 1e-6


def test_wide_and_delay_line_kernels_over_random_lengths(emu):
    """Track lengths and run lengths drawn at random (fixed seed) around the block boundaries of the two 16384-point
    kernels' small-plan twins: every length from 'shorter than a block' to 'a few blocks and a bit' must come out as
    the direct convolution -- the ends of a track are where the window arithmetic can be off by one."""
    rng = np.random.RandomState(2024)
    mid_side = lambda x: mo.mid_side(x.astype(np.float64))
    for taps, wide in ((512, True), (1024, True), (2048, False), (4096, False)):
        hop = 3 * taps if wide else taps // 2
        hm, hs = rng.randn(taps) / np.sqrt(taps), rng.randn(taps) / np.sqrt(taps)
        for _ in range(6):
            n = int(rng.choice([rng.randint(1, hop), hop * rng.randint(1, 5) + rng.randint(-2, 3), rng.randint(1, 6 * hop)]))
            n = max(1, n)
            x = np.ascontiguousarray((0.3 * rng.randn(n, 2)).astype(np.float32))
            y = np.zeros((n, 2), dtype=np.float32)
            ymid = np.zeros(n, dtype=np.float32)
            if wide:
                rc = emu.emu_convolve_wide(_fp(x), ctypes.c_longlong(n), _dp(hm), _dp(hs), ctypes.c_int(taps),
                                           ctypes.c_double(1.0), _fp(y), _fp(ymid), None)
            else:
                rc = emu.emu_convolve_delay(_fp(x), ctypes.c_longlong(n), _dp(hm), _dp(hs), ctypes.c_int(taps),
                                            ctypes.c_double(1.0), _fp(y), _fp(ymid), None, ctypes.c_int(int(rng.randint(1, 6))))
            assert rc == 0
            m, sd = mid_side(x)
            want, want_mid = mo.convolve_same(m, hm, sd, hs)
            assert np.abs(y - want).max() <= 5e-6, (taps, wide, n)
            assert np.abs(ymid - want_mid).max() <= 5e-6, (taps, wide, n)


def test_convolution_identity(emu):
    # scipy "same" centring (match_frequencies.py:112): delta at (F-1)//2 is the identity
    rng = np.random.RandomState(5)
    f = 256
    x = (0.4 * rng.randn(3001, 2)).astype(np.float32)
    delta = np.zeros(f)
    delta[(f - 1) // 2] = 1.0
    y, ymid, _ = emu_convolve(emu, x, delta, delta)
    assert np.abs(y - x).max() <= 2e-6
    assert np.abs(ymid - 0.5 * (x[:, 0] + x[:, 1])).max() <= 2e-6


def _native_config(case_cfg):
    import matchering_amd as mg

    kw = dict(case_cfg)
    lim = kw.pop("limiter", None)
    if lim is not None:
        kw["limiter"] = mg.LimiterConfig(**lim)
    return mg.Config(**kw)


@pytest.mark.parametrize("name", ["hot_lowrate", "quiet_reference", "custom_limiter", "cd_default"])
def test_analysis_phases(emu, name):
    t, r = build_inputs(CASES[name])
    cfg = _native_config(CASES[name]["config"])
    ocfg = oracle_params(CASES[name]["config"])
    native = cfg.to_native()
    for x, is_ref in ((t, 0), (r, 1)):
        x32 = np.ascontiguousarray(x, dtype=np.float32)
        n = x32.shape[0]
        max_div = int(n / cfg.max_piece_size) + 1
        half = cfg.fft_size // 2
        peak, amp, match = ctypes.c_double(), ctypes.c_double(), ctypes.c_double()
        div, piece = ctypes.c_int(), ctypes.c_longlong()
        rms = np.zeros(max_div)
        loud = np.zeros(max_div, dtype=np.int32)
        avg_mid, avg_side = np.zeros(half + 1), np.zeros(half + 1)
        rc = emu.emu_analyze(_fp(x32), ctypes.c_longlong(n), ctypes.byref(native), is_ref, ctypes.byref(peak),
                             ctypes.byref(amp), ctypes.byref(match), ctypes.byref(div), ctypes.byref(piece),
                             _dp(rms), loud.ctypes.data_as(ctypes.POINTER(ctypes.c_int)), _dp(avg_mid),
                             _dp(avg_side))
        assert rc == 0
        x64 = x32.astype(np.float64)
        c = 1.0
        if is_ref:
            x64, c = mo.peak_normalize(x64, ocfg.threshold, ocfg.min_value, False)
        a = mo.analyze(x64, ocfg)
        assert div.value == a.divisions and piece.value == a.piece
        assert abs(amp.value - c) <= 1e-7
        assert np.array_equal(np.flatnonzero(loud[: a.divisions]), a.loud_idx)
        assert np.abs(rms[: a.divisions] / a.rmses - 1).max() <= 1e-6
        assert abs(match.value / a.match_rms - 1) <= 1e-6
        for mine, rows in ((avg_mid, a.mid_loud), (avg_side, a.side_loud)):
            want = mo.average_spectrum(rows, ocfg.fft_size)
            assert np.abs(mine - want).max() <= 2e-6 * want.max()


def test_analysis_with_a_segment_of_two_transforms(emu):
    """fft_size 32768: one segment = two 16384-point transforms (AnalysisDouble: the real-FFT split
    R_k = A + w^k B) instead of one that would not fit a CU's LDS.  Levels and the averaged |rfft| of mid and
    side (match_frequencies.py:30-42) against the oracle, for a target and a reference with its
    normalisation."""
    import matchering_amd as mg
    from matchering_amd.synth import make_pair

    fft = 32768
    t, r = make_pair(7.0, 44100, pair=12, reference_seconds=5.0, reference_gain=0.6)
    cfg = mg.Config(fft_size=fft, max_piece_size=1.6)
    ocfg = mo.params(fft_size=fft, max_piece_size=1.6)
    native = cfg.to_native()
    for x32, is_ref in ((t, 0), (r, 1)):
        n = x32.shape[0]
        max_div = int(n / cfg.max_piece_size) + 1
        half = fft // 2
        peak, amp, match = ctypes.c_double(), ctypes.c_double(), ctypes.c_double()
        div, piece = ctypes.c_int(), ctypes.c_longlong()
        rms = np.zeros(max_div)
        loud = np.zeros(max_div, dtype=np.int32)
        avg_mid, avg_side = np.zeros(half + 1), np.zeros(half + 1)
        rc = emu.emu_analyze(_fp(x32), ctypes.c_longlong(n), ctypes.byref(native), is_ref, ctypes.byref(peak),
                             ctypes.byref(amp), ctypes.byref(match), ctypes.byref(div), ctypes.byref(piece),
                             _dp(rms), loud.ctypes.data_as(ctypes.POINTER(ctypes.c_int)), _dp(avg_mid),
                             _dp(avg_side))
        assert rc == 0
        x64 = x32.astype(np.float64)
        c = 1.0
        if is_ref:
            x64, c = mo.peak_normalize(x64, ocfg.threshold, ocfg.min_value, False)
        a = mo.analyze(x64, ocfg)
        assert div.value == a.divisions and piece.value == a.piece
        assert abs(amp.value - c) <= 1e-7
        assert np.array_equal(np.flatnonzero(loud[: a.divisions]), a.loud_idx)
        assert abs(match.value / a.match_rms - 1) <= 1e-6
        for mine, rows in ((avg_mid, a.mid_loud), (avg_side, a.side_loud)):
            want = mo.average_spectrum(rows, ocfg.fft_size)
            assert np.abs(mine - want).max() <= 2e-6 * want.max()


@pytest.mark.parametrize("log2h,seconds,piece", [(8, 0.5, 0.06), (10, 1.5, 0.25), (14, 9.0, 2.6)])
def test_analysis_with_a_segment_of_four_transforms(emu, log2h, seconds, piece):
    """fft_size = 4 N' (65536 with the 16384-point transform; AnalysisQuad): the real-FFT split applied twice, the even
    samples' spectrum through a scratch row, magnitudes added to the spectrum rows in place.  Levels and the averaged
    |rfft| of mid and side (match_frequencies.py:30-42) against the oracle, target and normalised reference; the small
    transforms run the same code in a fraction of the time."""
    import matchering_amd as mg
    from matchering_amd.synth import make_pair

    fft = 4 << log2h
    t, r = make_pair(seconds, 44100, pair=21, reference_seconds=seconds * 0.8, reference_gain=0.6)
    cfg = mg.Config(fft_size=fft, max_piece_size=piece)
    ocfg = mo.params(fft_size=fft, max_piece_size=piece)
    native = cfg.to_native()
    for x32, is_ref in ((t, 0), (r, 1)):
        n = x32.shape[0]
        max_div = int(n / cfg.max_piece_size) + 1
        half = fft // 2
        peak, amp, match = ctypes.c_double(), ctypes.c_double(), ctypes.c_double()
        div, piece_frames = ctypes.c_int(), ctypes.c_longlong()
        rms = np.zeros(max_div)
        loud = np.zeros(max_div, dtype=np.int32)
        avg_mid, avg_side = np.zeros(half + 1), np.zeros(half + 1)
        rc = emu.emu_analyze_quad(_fp(x32), ctypes.c_longlong(n), ctypes.byref(native), is_ref, ctypes.byref(peak),
                                  ctypes.byref(amp), ctypes.byref(match), ctypes.byref(div), ctypes.byref(piece_frames),
                                  _dp(rms), loud.ctypes.data_as(ctypes.POINTER(ctypes.c_int)), _dp(avg_mid),
                                  _dp(avg_side), ctypes.c_int(log2h))
        assert rc == 0
        x64 = x32.astype(np.float64)
        c = 1.0
        if is_ref:
            x64, c = mo.peak_normalize(x64, ocfg.threshold, ocfg.min_value, False)
        a = mo.analyze(x64, ocfg)
        assert div.value == a.divisions and piece_frames.value == a.piece
        assert abs(amp.value - c) <= 1e-7
        assert abs(peak.value - np.abs(x32).max()) <= 1e-7
        assert np.array_equal(np.flatnonzero(loud[: a.divisions]), a.loud_idx)
        assert np.abs(rms[: a.divisions] / a.rmses - 1).max() <= 1e-6
        assert abs(match.value / a.match_rms - 1) <= 1e-6
        for mine, rows in ((avg_mid, a.mid_loud), (avg_side, a.side_loud)):
            want = mo.average_spectrum(rows, ocfg.fft_size)
            assert np.abs(mine - want).max() <= 2e-6 * want.max()


@pytest.mark.parametrize("name", ["hot_lowrate", "custom_limiter"])
def test_limiter_phases(emu, name):
    t, r = build_inputs(CASES[name])
    ocfg = oracle_params(CASES[name]["config"])
    cfg = _native_config(CASES[name]["config"])
    native = cfg.to_native()
    tr = {}
    mo.master(t, r, ocfg, True, True, False, trace=tr)
    y = np.ascontiguousarray(tr["result_no_limiter"], dtype=np.float32)
    out = np.zeros_like(y)
    rc = emu.emu_limit(_fp(y), ctypes.c_longlong(y.shape[0]), ctypes.byref(native), ctypes.c_double(1.0),
                       ctypes.c_double(0.9), _fp(out), None, None)
    assert rc == 0
    want = mo.limit(y.astype(np.float64), ocfg) * 0.9
    assert rms_error(out, want) <= 1e-6
    assert np.abs(out - want).max() <= 5e-6


# attack / hold times at the edges of the chunk geometry: windows of a few samples (read out frame by
# frame), windows just over and under the 15-frame split, halos that need the 1024-block chunks
ODD_LIMITERS = [
    dict(sr=44100, attack=0.1, hold=0.1),          # 4-sample attack (window 9), 4-sample hold
    dict(sr=8000, attack=0.3, hold=0.5),           # attack 2 samples -> window 5; hold 4
    dict(sr=8000, attack=1.0, hold=1.0),           # attack 8 -> half window 8: the split form at its shortest
    dict(sr=44100, attack=0.18, hold=3.0),         # short attack window inside a long hold window
    dict(sr=44100, attack=8.0, hold=2.0),          # halos of ~220 blocks: 1024-block chunks
    dict(sr=96000, attack=3.0, hold=12.0),         # the same at 96 kHz, long hold
]


@pytest.mark.parametrize("lim", ODD_LIMITERS)
def test_limiter_unusual_attack_and_hold_times(emu, lim):
    import matchering_amd as mg
    from matchering_amd.synth import synth

    sr = lim["sr"]
    kw = dict(attack=lim["attack"], hold=lim["hold"])
    x = synth(1.2 if sr > 50000 else 2.0, sr, 9).astype(np.float64)
    x *= 1.5 / np.abs(x).max()
    y = np.ascontiguousarray(x, dtype=np.float32)
    cfg = mg.Config(internal_sample_rate=sr, limiter=mg.LimiterConfig(**kw))
    native = cfg.to_native()
    out = np.zeros_like(y)
    rc = emu.emu_limit(_fp(y), ctypes.c_longlong(y.shape[0]), ctypes.byref(native), ctypes.c_double(1.0),
                       ctypes.c_double(1.0), _fp(out), None, None)
    assert rc == 0
    want = mo.limit(y.astype(np.float64), mo.params(internal_sample_rate=sr, **kw))
    assert rms_error(out, want) <= 1e-6
    assert np.abs(out - want).max() <= 5e-6


def test_host_lowess_with_robustness_iterations(emu):
    """fir_design.cpp: lowess() against the compiled statsmodels' known answers for 1..3 robustness passes."""
    g = np.load(os.path.join(ROOT, "tests", "golden", "lowess_robust_kat.npz"))
    y = np.ascontiguousarray(g["y"], dtype=np.float64)
    for it in (1, 2, 3):
        fit = np.zeros_like(y)
        emu.emu_lowess_robust(_dp(y), ctypes.c_int(len(y)), ctypes.c_double(float(g["frac"])),
                              ctypes.c_double(float(g["delta"])), ctypes.c_int(it), _dp(fit))
        assert np.abs(fit - g[f"fit{it}"]).max() <= 1e-11


@pytest.mark.parametrize("it", [1, 3])
def test_fir_design_phases_with_lowess_iterations(emu, it):
    """The device phase functions of the FIR design (fir_plan.h) run on the host with Config.lowess_it > 0
    -- per-anchor weighted regressions with robustness weights, residuals, median, bisquare -- against the
    oracle's design (match_frequencies.py:45-101)."""
    import matchering_amd as mg

    fft = 1024
    rng = np.random.RandomState(3)
    bins = fft // 2 + 1
    k = np.arange(bins)
    a_t = (1.0 / (1.0 + k / 40.0) + 0.02 * rng.rand(bins)) * fft
    a_r = (1.2 / (1.0 + k / 25.0) + 0.02 * rng.rand(bins)) * fft
    a_r[100] *= 8.0                                  # a resonance the robust passes discount
    a_r[300:304] *= 0.1
    cfg = mg.Config(fft_size=fft, lowess_it=it)
    native = cfg.to_native()
    taps, raw, smooth = np.zeros(fft), np.zeros(bins), np.zeros(bins)
    assert emu.emu_design_fir(ctypes.byref(native), _dp(a_t), _dp(a_r), _dp(taps), _dp(raw), _dp(smooth)) == 0
    p = mo.params(fft_size=fft, lowess_it=it)
    want_smooth = mo.smooth_matching_curve(a_r / np.maximum(p.min_value, a_t), p)
    assert np.abs(smooth - want_smooth).max() <= 1e-9 * np.abs(want_smooth).max()
    p0 = mo.params(fft_size=fft)
    assert np.abs(want_smooth - mo.smooth_matching_curve(a_r / np.maximum(p.min_value, a_t), p0)).max() > 1e-3


@pytest.mark.parametrize("fft,sr", [(256, 44100), (1024, 96000)])
def test_fir_operator_in_two_factors_through_the_lowess_anchors(emu, fft, sr):
    """raw -> smooth as B (A raw) (mgx.hip build_fir_factors; k_fir_apply_a / k_fir_apply_b), built and applied on the
    host with the device's own phase functions: equal to the chain run on the curve itself (emu_design_fir) and to the
    oracle's smoothing (match_frequencies.py:45-75) to float64 rounding, bins 0 and 1 pinned, and both factors banded
    (their windows a small part of anchors x bins)."""
    import matchering_amd as mg

    rng = np.random.RandomState(fft)
    bins = fft // 2 + 1
    k = np.arange(bins)
    a_t = (1.0 / (1.0 + k / 40.0) + 0.02 * rng.rand(bins)) * fft
    a_r = (1.2 / (1.0 + k / 25.0) + 0.02 * rng.rand(bins)) * fft
    a_r[bins // 3] *= 6.0
    cfg = mg.Config(fft_size=fft, internal_sample_rate=sr, max_piece_size=fft * 4 / sr)
    native = cfg.to_native()
    taps, raw, chain = np.zeros(fft), np.zeros(bins), np.zeros(bins)
    assert emu.emu_design_fir(ctypes.byref(native), _dp(a_t), _dp(a_r), _dp(taps), _dp(raw), _dp(chain)) == 0
    factored = np.zeros(bins)
    a_size, b_size = ctypes.c_longlong(), ctypes.c_longlong()
    emu.emu_fir_factored.argtypes = [ctypes.c_void_p, c_double_p, c_double_p, ctypes.POINTER(ctypes.c_longlong),
                                     ctypes.POINTER(ctypes.c_longlong)]
    assert emu.emu_fir_factored(ctypes.byref(native), _dp(raw), _dp(factored), ctypes.byref(a_size),
                                ctypes.byref(b_size)) == 0
    scale = np.abs(chain).max()
    assert np.abs(factored - chain).max() <= 1e-12 * scale
    assert factored[0] == 0.0 and factored[1] == raw[1]
    p = mo.params(fft_size=fft, internal_sample_rate=sr)
    want = mo.smooth_matching_curve(a_r / np.maximum(p.min_value, a_t), p)
    assert np.abs(factored - want).max() <= 1e-9 * np.abs(want).max()
    anchors = emu.emu_fir_anchors(ctypes.byref(native))
    assert 0 < anchors <= (fft // 2) * cfg.lin_log_oversampling + 1
    assert 0 < a_size.value < 0.5 * anchors * bins and 0 < b_size.value < 0.2 * anchors * bins


def test_butterworth_design_matches_scipy(emu):
    """butter_tf (host_params.h) against scipy.signal.butter for the orders and cut-offs the limiter uses
    (hyrax.py:55-72): 7 Hz hold, 800 / 3000 Hz release, sample rates 8 k to 192 k."""
    from scipy import signal

    emu.emu_butter.argtypes = [ctypes.c_int, ctypes.c_double, ctypes.c_double, c_double_p, c_double_p]
    for order in (1, 2, 3, 4):          # (the design itself is good at any order)
        for fc, fs in ((7.0, 44100), (800.0 / 3000.0, 44100), (7.0, 8000), (800.0 / 3000.0, 192000), (50.0, 48000)):
            b, a = np.zeros(order + 1), np.zeros(order + 1)
            emu.emu_butter(order, fc, float(fs), _dp(b), _dp(a))
            wb, wa = signal.butter(order, fc, fs=fs)
            assert np.allclose(a, wa, rtol=0, atol=1e-13), (order, fc, fs, a, wa)
            assert np.allclose(b, wb, rtol=1e-9, atol=0), (order, fc, fs, b, wb)


# (hold order, release order): the general kernel's K is the larger of the two
FILTER_ORDERS = [(2, 2), (1, 2), (2, 1), (3, 1), (3, 2)]


@pytest.mark.parametrize("orders", FILTER_ORDERS)
def test_limiter_hold_and_release_filters_of_higher_order(emu, orders):
    """hold_filter_order / release_filter_order > 1 (defaults.py:39-47, hyrax.py:55-72): K-state sections,
    matrix-valued block maps and look-back (limiter_general.h), 10 s = ~125 chunks at 44.1 kHz."""
    import matchering_amd as mg
    from matchering_amd.synth import synth

    sr = 44100
    kw = dict(hold_filter_order=orders[0], release_filter_order=orders[1])
    rng = np.random.RandomState(13)
    x = synth(10.0, sr, 7).astype(np.float64)
    x *= 1.7 / np.abs(x).max()
    x[: sr] *= 0.3
    x += 1e-3 * rng.randn(*x.shape)
    y = np.ascontiguousarray(x, dtype=np.float32)
    cfg = mg.Config(internal_sample_rate=sr, limiter=mg.LimiterConfig(**kw))
    native = cfg.to_native()
    out = np.zeros_like(y)
    rc = emu.emu_limit(_fp(y), ctypes.c_longlong(y.shape[0]), ctypes.byref(native), ctypes.c_double(1.0),
                       ctypes.c_double(1.0), _fp(out), None, None)
    assert rc == 0
    want = mo.limit(y.astype(np.float64), mo.params(internal_sample_rate=sr, **kw))
    assert rms_error(out, want) <= 1e-6, rms_error(out, want)
    assert np.abs(out - want).max() <= 1e-5


def test_third_order_release_filter_where_it_is_well_conditioned(emu):
    """At 8 kHz the default release filter sits at d = 2 pi 0.267 / 8000 = 2.1e-4 from z = 1: order 3 is estimated at 1.7e-7 of
    full scale, under the 1e-6 limit, so it runs -- and must then agree with the reference's recursion."""
    import matchering_amd as mg
    from matchering_amd.synth import synth

    sr = 8000
    kw = dict(release_filter_order=3)
    rng = np.random.RandomState(5)
    x = synth(30.0, sr, 3).astype(np.float64)
    x *= 1.7 / np.abs(x).max()
    x += 1e-3 * rng.randn(*x.shape)
    y = np.ascontiguousarray(x, dtype=np.float32)
    native = mg.Config(internal_sample_rate=sr, limiter=mg.LimiterConfig(**kw)).to_native()
    out = np.zeros_like(y)
    rc = emu.emu_limit(_fp(y), ctypes.c_longlong(y.shape[0]), ctypes.byref(native), ctypes.c_double(1.0),
                       ctypes.c_double(1.0), _fp(out), None, None)
    assert rc == 0
    want = mo.limit(y.astype(np.float64), mo.params(internal_sample_rate=sr, **kw))
    assert rms_error(out, want) <= 2e-6, rms_error(out, want)


def test_limiter_filters_are_refused_by_conditioning_not_by_order(emu):
    """host_params.h limiter_params: a filter is refused when the rounding noise of the reference's own float64 recursion,
    ~1.1e-16 / (2 pi fc / fs)^(order - 1/2), exceeds 1e-6 (tests/test_limiter_order3_conditioning.py measures it)."""
    import matchering_amd as mg

    y = np.zeros((4096, 2), np.float32)
    out = np.zeros_like(y)

    def rc(**kw):
        native = mg.Config(limiter=mg.LimiterConfig(**kw)).to_native()
        return emu.emu_limit(_fp(y), ctypes.c_longlong(y.shape[0]), ctypes.byref(native), ctypes.c_double(1.0),
                             ctypes.c_double(1.0), _fp(out), None, None)

    assert rc(hold_filter_order=3) == 0                                   # 7 Hz at order 3: ~4e-9
    assert rc(hold_filter_order=3, release_filter_order=2) == 0
    assert rc(release_filter_order=3) == -1                               # 0.27 Hz at order 3: ~1e-5
    assert rc(hold_filter_order=3, release_filter_order=3) == -1
    assert rc(hold_filter_order=4) == -1                                  # ~3e-6 at 7 Hz
    assert rc(hold_filter_order=4, hold_filter_coefficient=400.0) == -1   # clean, but no instantiation above 3
    assert rc(release_filter_order=3, release_filter_coefficient=20000.0) == 0     # a 6.7 Hz release filter is clean at 3


def test_limiter_lookback_across_many_chunks(emu):
    """Default 44.1 kHz limiter on 14 s of hot material: ~85 chunks, so the release filter's carry
    is a truncated look-back over ~70 predecessor chunks and the hold filter's over 3."""
    import matchering_amd as mg
    from matchering_amd.synth import synth

    sr = 44100
    rng = np.random.RandomState(11)
    x = synth(14.0, sr, 5).astype(np.float64)
    x *= 1.6 / np.abs(x).max()
    x[: sr // 2] *= 0.2                     # quiet start: the limiter engages mid-way
    x += 1e-3 * rng.randn(*x.shape)
    y = np.ascontiguousarray(x, dtype=np.float32)
    cfg = mg.Config()
    native = cfg.to_native()
    out = np.zeros_like(y)
    rc = emu.emu_limit(_fp(y), ctypes.c_longlong(y.shape[0]), ctypes.byref(native), ctypes.c_double(1.0),
                       ctypes.c_double(1.0), _fp(out), None, None)
    assert rc == 0
    ocfg = mo.params()
    want = mo.limit(y.astype(np.float64), ocfg)
    assert np.abs(want).max() <= ocfg.threshold * (1 + 1e-9)
    assert rms_error(out, want) <= 1e-6
    assert np.abs(out - want).max() <= 5e-6


@pytest.mark.parametrize("n", [8, 23, 3536, 3537, 2 * 3536 - 1, 7072, 30011])
def test_limiter_lengths_around_chunk_edges(emu, n):
    import matchering_amd as mg

    rng = np.random.RandomState(n)
    y = (0.9 * rng.randn(n, 2)).astype(np.float32)
    cfg = mg.Config()
    native = cfg.to_native()
    out = np.zeros_like(y)
    rc = emu.emu_limit(_fp(y), ctypes.c_longlong(n), ctypes.byref(native), ctypes.c_double(1.0),
                       ctypes.c_double(1.0), _fp(out), None, None)
    assert rc == 0
    want = mo.limit(y.astype(np.float64), mo.params())
    assert np.abs(out - want).max() <= 5e-6


def test_quiet_chunk_path_applies_to_the_default_filters_and_not_to_close_poles(emu):
    """limiter_fill (host_params.h) lets the closed-form quiet chunks run only where their float32 difference of two
    exponentials is good to 1e-7 of the hold carry: |alpha_release - alpha_hold| > 4 beta_release.  The default
    filters (7 Hz hold, 0.27 Hz release) are 25 beta apart and must qualify -- a stricter bound once switched the path
    off for every default call, which only the HBM counters noticed (448 instead of 392 MB per limiter launch)."""
    import matchering_amd as mg

    assert emu.emu_limiter_quiet_ok(ctypes.byref(mg.Config().to_native())) == 1
    assert emu.emu_limiter_quiet_ok(ctypes.byref(mg.Config(internal_sample_rate=96000).to_native())) == 1
    # release pole moved onto the hold pole: 0.27 Hz * 26 = 7 Hz
    close = mg.Config(limiter=mg.LimiterConfig(release_filter_coefficient=7.0 * 3000.0 * 1.0001))
    assert emu.emu_limiter_quiet_ok(ctypes.byref(close.to_native())) == 0
