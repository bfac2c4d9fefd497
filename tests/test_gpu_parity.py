"""GPU parity: the HIP path (through the C ABI) against the oracle and against
the outputs frozen from the unmodified reference (tests/golden).

Tolerance (BASELINE.json north_star): <= 1e-5 RMS on float32 against the float64
numpy reference, absolute with full scale = 1.0.  Measured errors are ~1e-7; the
assertions keep the stated 1e-5 for the final outputs and use tighter, documented
bounds where a stage is checked on its own.
"""

import os

import numpy as np
import pytest

import mastering_oracle as mo
from cases import CASES, build_inputs, oracle_params
from conftest import rms_error

pytestmark = pytest.mark.gpu

RMS_TOL = 1e-5


def make_config(case_cfg):
    import matchering_amd as mg

    kw = dict(case_cfg)
    lim = kw.pop("limiter", None)
    if lim is not None:
        kw["limiter"] = mg.LimiterConfig(**lim)
    return mg.Config(**kw)


@pytest.fixture(scope="module")
def oracle_runs():
    cache = {}

    def run(name):
        if name not in cache:
            t, r = build_inputs(CASES[name])
            tr = {}
            outs = mo.master(t, r, oracle_params(CASES[name]["config"]), True, True, True, trace=tr)
            cache[name] = (t, r, outs, tr)
        return cache[name]

    return run


@pytest.mark.parametrize("name", sorted(CASES))
def test_master_matches_reference_golden(name, golden, oracle_runs):
    from matchering_amd import stages

    g = golden(name)
    t, r, outs, tr = oracle_runs(name)
    cfg = make_config(CASES[name]["config"])
    res, res_nl, res_nln = stages.main(t, r, cfg, need_default=True, need_no_limiter=True,
                                       need_no_limiter_normalized=True)
    assert res.dtype == np.float32 and res.shape == t.shape
    # against the frozen reference outputs
    assert rms_error(res, g["result_f32"]) <= RMS_TOL
    assert rms_error(res_nl, g["result_no_limiter_f32"]) <= RMS_TOL
    idx = g["sparse_index"]
    assert rms_error(res_nln[idx], g["result_no_limiter_normalized_sparse"]) <= RMS_TOL
    # and against the oracle run here
    for mine, want in zip((res, res_nl, res_nln), outs):
        assert rms_error(mine, want) <= RMS_TOL
        assert np.abs(mine - want).max() <= 2e-5 * max(1.0, np.abs(want).max())
    # brick-wall property (hyrax.py:87,97,99)
    assert np.abs(res).max() <= cfg.threshold * tr["final_amplitude_coefficient"] * (1 + 1e-6)


@pytest.mark.parametrize("name", sorted(CASES))
def test_fir_and_stage_scalars_match_reference_golden(name, golden):
    """What flows BETWEEN the stages of stages.main, device values against the values frozen from the
    unmodified reference: the matching FIR pair (match_frequencies.py:78-101, read back through
    mgx_last_fir) and every scalar of mgx_report (stages.py:80-91, 149-168, 186-191;
    match_levels.py:29-44).  An end-to-end comparison alone would let a FIR level error hide
    behind a compensating gain."""
    import ctypes

    from matchering_amd._native import check, library
    from matchering_amd.device import default_device

    g = golden(name)
    t, r = build_inputs(CASES[name])
    cfg = make_config(CASES[name]["config"])
    dev = default_device()
    with dev.lock:
        td, rd = dev.upload(t), dev.upload(r)
        outs = [dev.alloc(t.shape[0] * 8) for _ in range(3)]
        try:
            rep = dev.master(td, t.shape[0], rd, r.shape[0], cfg.to_native(), *outs)
            taps_dev, taps = ctypes.c_void_p(), ctypes.c_int32()
            check(library().mgx_last_fir(dev.handle, ctypes.byref(taps_dev), ctypes.byref(taps)))
            assert taps.value == cfg.fft_size
            fir = dev.download(int(taps_dev.value), (2, taps.value))
        finally:
            for b in (td, rd, *outs):
                b.release()
    for mine, want in ((fir[0], g["fir_mid"]), (fir[1], g["fir_side"])):
        assert np.abs(mine - want).max() <= 1e-6 * np.abs(want).max()          # float32 taps: 6e-8 of the peak tap each
    rel = lambda a, b: abs(a / b - 1.0)                                          # noqa: E731
    assert rel(rep.rms_coefficient, float(g["rms_coefficient"])) <= 1e-6
    assert rel(rep.final_amplitude_coefficient, float(g["final_amplitude_coefficient"])) <= 1e-6
    assert rel(rep.target_match_rms, float(g["target_match_rms"])) <= 1e-6
    assert rel(rep.reference_match_rms, float(g["reference_match_rms"])) <= 1e-6
    steps = cfg.rms_correction_steps
    got = np.array(rep.correction_coefficients[:steps])
    assert got.shape == g["correction_coefficients"].shape
    assert np.abs(got / g["correction_coefficients"] - 1.0).max() <= 1e-6
    assert rel(rep.normalize_coefficient, float(g["normalize_coefficient"])) <= 1e-6
    assert (rep.target_divisions, rep.reference_divisions) == (int(g["target_divisions"]), int(g["reference_divisions"]))
    assert (rep.target_piece, rep.reference_piece) == (int(g["target_piece"]), int(g["reference_piece"]))
    assert (rep.target_loud_count, rep.reference_loud_count) == (int(g["target_loud_count"]), int(g["reference_loud_count"]))
    assert bool(rep.limiter_active) == bool(g["limiter_active"])


@pytest.mark.parametrize("name", sorted(CASES))
def test_analysis_stage(name, oracle_runs):
    from matchering_amd import kernels

    t, r, _, tr = oracle_runs(name)
    cfg = make_config(CASES[name]["config"])
    st = kernels.analyze(t, cfg, is_reference=False)
    assert st.divisions == tr["target_divisions"] and st.piece_size == tr["target_piece"]
    assert np.array_equal(np.flatnonzero(st.loud), tr["target_loud_idx"])
    assert np.abs(st.rmses / tr["target_rmses"] - 1).max() <= 1e-7
    assert abs(st.match_rms / tr["target_match_rms"] - 1) <= 1e-7
    c0 = tr["rms_coefficient"]
    # float32 FFT against float64: relative error of a bin vs the spectrum's peak level
    for mine, want in ((st.average_spectrum_mid * c0, tr["mid"].avg_target),
                       (st.average_spectrum_side * c0, tr["side"].avg_target)):
        # (a mono target's side spectrum is exactly zero in float64 and rounding noise of the two-for-one
        # transform here, ~1e-8 of the mid level: far below min_value = 1e-6, where match_frequencies.py:53
        # floors it)
        assert np.abs(mine - want).max() <= max(2e-6 * want.max(), 1e-7)
    sr = kernels.analyze(r, cfg, is_reference=True)
    assert abs(sr.amplitude_coefficient - tr["final_amplitude_coefficient"]) <= 1e-7
    assert np.array_equal(np.flatnonzero(sr.loud), tr["reference_loud_idx"])
    assert abs(sr.match_rms / tr["reference_match_rms"] - 1) <= 1e-7
    for mine, want in ((sr.average_spectrum_mid, tr["mid"].avg_reference),
                       (sr.average_spectrum_side, tr["side"].avg_reference)):
        assert np.abs(mine - want).max() <= 2e-6 * want.max()


@pytest.mark.parametrize("taps", [64, 256, 1024, 4096, 8192])
def test_convolution_stage(taps):
    from matchering_amd import kernels

    rng = np.random.RandomState(taps)
    n = 3 * taps + 1237
    x = (0.3 * rng.randn(n, 2)).astype(np.float32)
    hm, hs = rng.randn(taps) / np.sqrt(taps), rng.randn(taps) / np.sqrt(taps)
    y, ymid, peak = kernels.convolve(x, hm, hs, gain=1.7)
    mid, side = mo.mid_side(x.astype(np.float64))
    want, want_mid = mo.convolve_same(mid * 1.7, hm, side * 1.7, hs)
    assert rms_error(y, want) <= 1e-6
    assert rms_error(ymid, want_mid) <= 1e-6
    assert abs(peak - np.abs(y).max()) <= 1e-6


@pytest.mark.parametrize("taps", [16384, 32768, 65536])
def test_partitioned_convolution_stage(taps):
    """Uniformly partitioned overlap-save: FIRs longer than half an LDS block (a 16 k-tap filter on
    8192-frame blocks = 4 partitions; BASELINE config #5)."""
    from matchering_amd import kernels

    rng = np.random.RandomState(taps)
    n = 2 * taps + 40961
    x = (0.3 * rng.randn(n, 2)).astype(np.float32)
    hm, hs = rng.randn(taps) / np.sqrt(taps), rng.randn(taps) / np.sqrt(taps)
    y, ymid, peak = kernels.convolve(x, hm, hs, gain=0.9)
    mid, side = mo.mid_side(x.astype(np.float64))
    want, want_mid = mo.convolve_same(mid * 0.9, hm, side * 0.9, hs)
    assert rms_error(y, want) <= 1e-6
    assert rms_error(ymid, want_mid) <= 1e-6
    assert abs(peak - np.abs(y).max()) <= 1e-6


@pytest.mark.parametrize("n", [1, 8191, 8192, 3 * 8192, 300 * 8192 + 4321, 1500 * 8192 + 17])
def test_two_partition_filter_as_a_delay_line(n, monkeypatch):
    """16384 taps on 16384-point blocks (config #5): one transform per block for both channels, partition 1's
    product carried from block to block of a workgroup's run (k_conv_delay).  Track lengths from one frame to runs of
    six blocks per workgroup, against fftconvolve and against the partitioned kernel it replaces."""
    from matchering_amd import kernels

    taps = 16384
    rng = np.random.RandomState(n % 100000)
    x = (0.3 * rng.randn(n, 2)).astype(np.float32)
    hm, hs = rng.randn(taps) / np.sqrt(taps), rng.randn(taps) / np.sqrt(taps)
    y, ymid, peak = kernels.convolve(x, hm, hs, gain=0.9)
    mid, side = mo.mid_side(x.astype(np.float64))
    want, want_mid = mo.convolve_same(mid * 0.9, hm, side * 0.9, hs)
    assert rms_error(y, want) <= 1e-6
    assert rms_error(ymid, want_mid) <= 1e-6
    assert abs(peak - np.abs(y).max()) <= 1e-6
    monkeypatch.setenv("MGX_NO_CONV_DELAY", "1")
    y2, ymid2, peak2 = kernels.convolve(x, hm, hs, gain=0.9)
    assert np.abs(y - y2).max() <= 5e-6 and np.abs(ymid - ymid2).max() <= 5e-6 and abs(peak - peak2) <= 5e-6


@pytest.mark.parametrize("n", [1, 12287, 12288, 12289, 5 * 12288 + 777, 300 * 12288 + 4321])
def test_default_filter_on_wide_blocks(n, monkeypatch):
    """4096 taps (the reference's default fft_size) on 16384-point blocks, three quarters of a block fresh output
    (k_conv_wide): from one frame to more blocks than workgroups, against fftconvolve and against the N = 2F kernel."""
    from matchering_amd import kernels

    taps = 4096
    rng = np.random.RandomState(n % 100000)
    x = (0.3 * rng.randn(n, 2)).astype(np.float32)
    hm, hs = rng.randn(taps) / np.sqrt(taps), rng.randn(taps) / np.sqrt(taps)
    y, ymid, peak = kernels.convolve(x, hm, hs, gain=1.1)
    mid, side = mo.mid_side(x.astype(np.float64))
    want, want_mid = mo.convolve_same(mid * 1.1, hm, side * 1.1, hs)
    assert rms_error(y, want) <= 1e-6
    assert rms_error(ymid, want_mid) <= 1e-6
    assert abs(peak - np.abs(y).max()) <= 1e-6
    monkeypatch.setenv("MGX_NO_CONV_WIDE", "1")
    y2, ymid2, peak2 = kernels.convolve(x, hm, hs, gain=1.1)
    assert np.abs(y - y2).max() <= 5e-6 and np.abs(ymid - ymid2).max() <= 5e-6 and abs(peak - peak2) <= 5e-6


def test_master_long_fir_96k():
    """BASELINE config #5 in miniature: 96 kHz, fft_size 16384 (16 k-tap matching FIR), full pipeline
    against the oracle."""
    import matchering_amd as mg
    from matchering_amd import stages
    from matchering_amd.synth import make_pair

    sr = 96000
    t, r = make_pair(3.0, sr, pair=7, reference_seconds=2.6)
    kw = dict(internal_sample_rate=sr, fft_size=16384, max_piece_size=1.1)
    res, res_nl, res_nln = stages.main(t, r, mg.Config(**kw), need_default=True, need_no_limiter=True,
                                       need_no_limiter_normalized=True)
    want = mo.master(t, r, mo.params(**kw), True, True, True)
    for mine, ref in zip((res, res_nl, res_nln), want):
        assert rms_error(mine, ref) <= RMS_TOL
        assert np.abs(mine - ref).max() <= 2e-5 * max(1.0, np.abs(ref).max())


def _master_with_taps(t, r, cfg):
    """stages.main's three outputs through the device API, plus the FIR pair the call designed (mgx_last_fir)."""
    import ctypes

    from matchering_amd._native import check, library
    from matchering_amd.device import default_device

    dev = default_device()
    with dev.lock:
        td, rd = dev.upload(t), dev.upload(r)
        outs = [dev.alloc(t.shape[0] * 8) for _ in range(3)]
        try:
            dev.master(td, t.shape[0], rd, r.shape[0], cfg.to_native(), *outs)
            taps_dev, taps = ctypes.c_void_p(), ctypes.c_int32()
            check(library().mgx_last_fir(dev.handle, ctypes.byref(taps_dev), ctypes.byref(taps)))
            assert taps.value == cfg.fft_size
            fir = dev.download(int(taps_dev.value), (2, taps.value))
            res = [np.array(dev.download(o, (t.shape[0], 2))) for o in outs]
            fir = np.array(fir)
        finally:
            for b in (td, rd, *outs):
                b.release()
    return res, fir


@pytest.mark.parametrize("fft", [32768, 65536])
def test_master_fft_size_32768_and_65536(fft):
    """The two sizes above what one transform in a CU's LDS holds: analysis segments of two / four 16384-point
    transforms (k_analyze_double, k_analyze_quad), the raw -> smooth operator in its two packed factors, the taps by
    8 / 16 sub-transforms, 32 k / 64 k taps convolved in four / eight partitions.  192 kHz, where such sizes are at
    home: the FIR pair and the full pipeline against the oracle; fft_size 131072 must be refused."""
    import matchering_amd as mg
    from matchering_amd import stages
    from matchering_amd._native import MgxError
    from matchering_amd.synth import make_pair

    sr = 192000
    t, r = make_pair(2.0, sr, pair=14, reference_seconds=1.7)
    kw = dict(internal_sample_rate=sr, fft_size=fft, max_piece_size=0.6)
    res, res_nl, res_nln = stages.main(t, r, mg.Config(**kw), need_default=True, need_no_limiter=True,
                                       need_no_limiter_normalized=True)
    tr = {}
    want = mo.master(t, r, mo.params(**kw), True, True, True, trace=tr)
    for mine, ref in zip((res, res_nl, res_nln), want):
        assert rms_error(mine, ref) <= RMS_TOL
        assert np.abs(mine - ref).max() <= 2e-5 * max(1.0, np.abs(ref).max())
    _, fir = _master_with_taps(t, r, mg.Config(**kw))
    for mine, ref in ((fir[0], tr["fir_mid"]), (fir[1], tr["fir_side"])):
        assert np.abs(mine - ref).max() <= 2e-6 * np.abs(ref).max()
    with pytest.raises(MgxError, match="fft_size"):
        stages.main(t, r, mg.Config(internal_sample_rate=sr, fft_size=131072, max_piece_size=0.9))


@pytest.mark.parametrize("fft", [16384, 32768])
def test_factored_fir_operator_equals_the_round_4_paths(fft, monkeypatch):
    """fft_size >= 16384: raw -> smooth as B (A raw) through the anchors' LOWESS fits (two packed banded factors, a
    few MB) against what it replaces -- the dense 537 MB operator at 16384, the chain run on the curve itself at
    32768 (MGX_FIR_ROUND4=1).  The same linear map summed in another order: taps equal to float32 rounding."""
    import matchering_amd as mg
    from matchering_amd.synth import make_pair

    sr = 96000
    t, r = make_pair(3.0, sr, pair=9, reference_seconds=2.6)
    cfg = mg.Config(internal_sample_rate=sr, fft_size=fft, max_piece_size=1.1)
    res, fir = _master_with_taps(t, r, cfg)
    monkeypatch.setenv("MGX_FIR_ROUND4", "1")
    res4, fir4 = _master_with_taps(t, r, cfg)
    assert np.abs(fir - fir4).max() <= 2e-7 * np.abs(fir4).max()
    for a, b in zip(res, res4):
        assert np.abs(a - b).max() <= 2e-6


def test_lowess_delta_zero_does_not_build_a_dense_factor():
    """with lowess_delta = 0 every point of the log grid is a LOWESS anchor, and the factored operator's dense
    intermediate (anchors x bins doubles) would be 2.1 GB at fft_size 16384 and 8.6 GB at 32768.  Above a 1 GiB budget the
    plan takes round 4's paths (the dense operator at 16384), and the FIR still equals the oracle's."""
    import matchering_amd as mg
    from matchering_amd.synth import make_pair

    sr = 96000
    t, r = make_pair(3.0, sr, pair=4, reference_seconds=2.6)
    kw = dict(internal_sample_rate=sr, fft_size=16384, max_piece_size=1.1, lowess_delta=0.0)
    res, fir = _master_with_taps(t, r, mg.Config(**kw))
    tr = {}
    want = mo.master(t, r, mo.params(**kw), True, True, True, trace=tr)
    ref = np.stack([tr["fir_mid"], tr["fir_side"]])
    assert np.abs(fir - ref).max() <= 2e-6 * np.abs(ref).max()
    assert rms_error(res[0], want[0]) <= RMS_TOL


def test_convolution_identity_and_linearity():
    from matchering_amd import kernels

    rng = np.random.RandomState(5)
    f = 1024
    x = (0.4 * rng.randn(50001, 2)).astype(np.float32)
    delta = np.zeros(f)
    delta[(f - 1) // 2] = 1.0          # scipy "same" centring: identity (match_frequencies.py:112)
    y, ymid, _ = kernels.convolve(x, delta, delta)
    assert np.abs(y - x).max() <= 2e-6
    assert np.abs(ymid - 0.5 * (x[:, 0] + x[:, 1])).max() <= 2e-6
    h1, h2 = rng.randn(f) / 32, rng.randn(f) / 32
    ya, _, _ = kernels.convolve(x, h1, h2)
    yb, _, _ = kernels.convolve(x, 2 * h1, 2 * h2)
    assert rms_error(2 * ya, yb) <= 1e-6


@pytest.mark.parametrize("name", sorted(CASES))
def test_limiter_stage(name, oracle_runs):
    from matchering_amd import kernels

    _, _, _, tr = oracle_runs(name)
    cfg = make_config(CASES[name]["config"])
    ocfg = oracle_params(CASES[name]["config"])
    y = tr["result_no_limiter"].astype(np.float32)
    out, active = kernels.limit(y, cfg, gain=1.0, post_gain=0.9)
    env = mo.limiter_envelopes(y.astype(np.float64), ocfg)
    want = mo.limit(y.astype(np.float64), ocfg) * 0.9
    assert active == (env is not None)
    assert rms_error(out, want) <= 1e-6
    assert np.abs(out - want).max() <= 5e-6
    # the filtfilt edges are exact, not approximated by a warm-up
    assert np.abs(out[:64] - want[:64]).max() <= 1e-6 and np.abs(out[-64:] - want[-64:]).max() <= 1e-6


@pytest.mark.parametrize("lim,decay_window", [(dict(), (9.0, 14.0)), (dict(attack=8.0, hold=2.0), (9.0, 14.0)),
                                              (dict(release=300.0, hold_filter_coefficient=40.0), (7.75, 8.5))])
def test_limiter_quiet_chunks_between_loud_ones(lim, decay_window):
    """Chunks without a frame above the threshold take limit_chunk_quiet (closed-form envelopes from the
    carries, no windows or scans): a track that is quiet for seconds between short bursts, so that the hold,
    release and attack states decay across many quiet chunks and are picked up again by busy ones -- against
    hyrax.py:78-99 restated, at the bounds of the other limiter tests.  Second case: 1024-block chunks;
    third: a fast release and a fast hold filter (large per-chunk decay)."""
    import matchering_amd as mg
    from matchering_amd import kernels
    from matchering_amd.synth import synth

    sr = 44100
    x = synth(20.0, sr, 31).astype(np.float64)
    x *= 0.55 / np.abs(x).max()                                   # nowhere near the threshold ...
    for start, length, level in ((1.0, 0.004, 1.4), (1.03, 0.05, 1.2), (7.5, 0.2, 1.7), (15.0, 0.0005, 1.3), (19.99, 0.005, 1.25)):
        a, b = int(start * sr), int((start + length) * sr)
        x[a:b] *= level / np.abs(x[a:b]).max()                     # ... except in five short bursts that peak at `level`
    y = x.astype(np.float32)
    assert 1e-4 < np.mean(np.abs(y).max(axis=1) > mg.Config().threshold) < 0.02
    out, active = kernels.limit(y, mg.Config(limiter=mg.LimiterConfig(**lim)), gain=1.0, post_gain=0.8)
    want = mo.limit(y.astype(np.float64), mo.params(**lim)) * 0.8
    assert active
    assert rms_error(out, want) <= 1e-6
    assert np.abs(out - want).max() <= 5e-6
    quiet = slice(int(decay_window[0] * sr), int(decay_window[1] * sr))   # after a burst: pure carry decay
    assert np.abs(out[quiet] - want[quiet]).max() <= 2e-6
    assert np.abs(want[quiet] - 0.8 * y[quiet]).max() > 1e-4        # (and the carries do matter there)


@pytest.mark.parametrize("lim", [
    dict(sr=44100, attack=0.1, hold=0.1),          # windows of a few samples, read out frame by frame
    dict(sr=8000, attack=1.0, hold=1.0),           # half window 8: the split form at its shortest
    dict(sr=44100, attack=0.18, hold=3.0),
    dict(sr=44100, attack=8.0, hold=2.0),          # halos of ~220 blocks: the 1024-block chunks
    dict(sr=96000, attack=3.0, hold=12.0),
])
def test_limiter_unusual_attack_and_hold_times(lim):
    """LimiterConfig accepts any positive attack / hold (defaults.py:39-46): the kernel's chunk geometry
    follows (limiter_kernel.h), checked here against hyrax.py:78-99 restated."""
    import matchering_amd as mg
    from matchering_amd import kernels
    from matchering_amd.synth import synth

    sr = lim["sr"]
    kw = dict(attack=lim["attack"], hold=lim["hold"])
    x = synth(3.0, sr, 9).astype(np.float64)
    x *= 1.5 / np.abs(x).max()
    y = x.astype(np.float32)
    out, active = kernels.limit(y, mg.Config(internal_sample_rate=sr, limiter=mg.LimiterConfig(**kw)))
    want = mo.limit(y.astype(np.float64), mo.params(internal_sample_rate=sr, **kw))
    assert active
    assert rms_error(out, want) <= 1e-6
    assert np.abs(out - want).max() <= 5e-6


@pytest.mark.parametrize("orders", [(2, 2), (1, 2), (2, 1)])
def test_limiter_hold_and_release_filters_of_order_two(orders):
    """hold_filter_order / release_filter_order = 2 (defaults.py:39-47, hyrax.py:55-72): k_limit_general,
    two-state sections with matrix-valued block maps and look-back (limiter_general.h).  20 s = 250 chunks,
    so the release carry is a sum over ~300 predecessors' state words."""
    import matchering_amd as mg
    from matchering_amd import kernels
    from matchering_amd.synth import synth

    sr = 44100
    kw = dict(hold_filter_order=orders[0], release_filter_order=orders[1])
    rng = np.random.RandomState(13)
    x = synth(20.0, sr, 7).astype(np.float64)
    x *= 1.7 / np.abs(x).max()
    x[:sr] *= 0.3
    x += 1e-3 * rng.randn(*x.shape)
    y = x.astype(np.float32)
    out, active = kernels.limit(y, mg.Config(internal_sample_rate=sr, limiter=mg.LimiterConfig(**kw)))
    want = mo.limit(y.astype(np.float64), mo.params(internal_sample_rate=sr, **kw))
    assert active
    assert rms_error(out, want) <= 1e-6
    assert np.abs(out - want).max() <= 1e-5


@pytest.mark.parametrize("orders", [(3, 1), (3, 2)])
def test_limiter_hold_filter_of_order_three(orders):
    """hold_filter_order = 3 (defaults.py:48-56 accepts any positive order, hyrax.py:61-66 runs
    it).  The 7 Hz hold filter is well conditioned at order 3 (its float64 recursion is good to 4e-9,
    tests/test_limiter_order3_conditioning.py); the release filter stays at an order whose conditioning passes.
    k_limit_general<3>: three-state sections, 3 x 3 block maps and look-back matrices."""
    import matchering_amd as mg
    from matchering_amd import kernels
    from matchering_amd.synth import synth

    sr = 44100
    kw = dict(hold_filter_order=orders[0], release_filter_order=orders[1])
    rng = np.random.RandomState(17)
    x = synth(20.0, sr, 9).astype(np.float64)
    x *= 1.7 / np.abs(x).max()
    x[:sr] *= 0.3
    x += 1e-3 * rng.randn(*x.shape)
    y = x.astype(np.float32)
    out, active = kernels.limit(y, mg.Config(internal_sample_rate=sr, limiter=mg.LimiterConfig(**kw)))
    want = mo.limit(y.astype(np.float64), mo.params(internal_sample_rate=sr, **kw))
    assert active
    assert rms_error(out, want) <= 1e-6
    assert np.abs(out - want).max() <= 1e-5


def test_master_with_third_order_hold_filter():
    """The whole path with LimiterConfig(hold_filter_order=3), against the oracle (1e-5 RMS, the project's bar)."""
    import matchering_amd as mg
    from matchering_amd import stages
    from matchering_amd.synth import make_pair

    kw = dict(hold_filter_order=3)
    target, reference = make_pair(12.0, 44100, pair=4, reference_seconds=9.0)
    got = stages.main(target, reference, mg.Config(max_piece_size=3.0, limiter=mg.LimiterConfig(**kw)))
    want = mo.master(target, reference, mo.params(max_piece_size=3.0, **kw), True, False, False)
    assert rms_error(got[0], want[0]) <= RMS_TOL


def test_master_with_second_order_limiter_filters():
    """The whole path with the limiter's filters at order 2, against the oracle."""
    import matchering_amd as mg
    from matchering_amd import stages
    from matchering_amd.synth import make_pair

    kw = dict(hold_filter_order=2, release_filter_order=2)
    target, reference = make_pair(12.0, 44100, pair=3, reference_seconds=9.0)
    got = stages.main(target, reference, mg.Config(max_piece_size=3.0, limiter=mg.LimiterConfig(**kw)))
    want = mo.master(target, reference, mo.params(max_piece_size=3.0, **kw), True, False, False)
    assert rms_error(got[0], want[0]) <= RMS_TOL


def test_clipped_piece_sumsq():
    from matchering_amd import kernels

    rng = np.random.RandomState(3)
    mid = (0.8 * rng.randn(100003)).astype(np.float32)
    piece, div = 9001, 11
    got = kernels.clipped_piece_sumsq(mid, piece, div, gain=1.3)
    rows = np.clip(mid[: piece * div].astype(np.float64) * 1.3, -1, 1).reshape(div, piece)
    want = np.einsum("ij,ij->i", rows, rows)
    assert np.abs(got / want - 1).max() <= 1e-12


def test_fails_loudly_on_unsupported():
    import matchering_amd as mg
    from matchering_amd import stages
    from matchering_amd._native import MgxError

    t, r = build_inputs(CASES["hot_lowrate"])
    with pytest.raises(MgxError):
        stages.main(t, r, mg.Config(internal_sample_rate=8000, fft_size=131072, max_piece_size=17.0,
                                    max_length=600))
    # fft_size 2 and 4 pass defaults.py:110-112 and then fail inside the reference (tests/test_host_vs_reference.py)
    with pytest.raises(MgxError, match="fft_size below 8"):
        stages.main(t, r, mg.Config(internal_sample_rate=8000, fft_size=4, max_piece_size=1.0))
    # a third-order RELEASE filter: ill-conditioned in the reference's own form (limiter_general.h) -- refused by its
    # conditioning, not by its order (a third-order hold filter runs: test_limiter_hold_filter_of_order_three)
    # (at 44.1 kHz: 1.2e-5 of full scale; at 8 kHz the same filter sits five times further from z = 1 and passes)
    with pytest.raises(MgxError, match="ill-conditioned"):
        stages.main(t, r, mg.Config(internal_sample_rate=44100, max_piece_size=5.0,
                                    limiter=mg.LimiterConfig(release_filter_order=3)))
    with pytest.raises(MgxError, match="orders up to 3"):          # clean at this cut-off, but nothing is built for order 4
        stages.main(t, r, mg.Config(internal_sample_rate=8000, max_piece_size=5.0,
                                    limiter=mg.LimiterConfig(hold_filter_order=4, hold_filter_coefficient=400.0)))


def test_process_files_end_to_end(tmp_path):
    """``mg.process`` (core.py:32-121): WAV files in, WAV files + previews out, against the oracle run
    on the decoded inputs.  BASELINE config #1 (examples/basic.py) with the DSP on the GPU."""
    import matchering_amd as mg
    from matchering_amd import audio_io
    from matchering_amd.synth import make_pair

    sr = 44100
    t, r = make_pair(9.0, sr, pair=9, reference_seconds=5.3)
    tp, rp = str(tmp_path / "target.wav"), str(tmp_path / "reference.wav")
    audio_io.write_wav(tp, t, sr, "PCM_24")
    audio_io.write_wav(rp, r[:, :1], sr, "FLOAT")                 # mono reference: 2201 + duplication
    outs = {k: str(tmp_path / f"{k}.wav") for k in ("lim24", "lim16", "nolim", "norm", "pt", "pr")}
    cfg = mg.Config(max_piece_size=1.0, preview_size=6, preview_analysis_step=1.5, preview_fade_size=0.5)
    codes = []
    mg.log(lambda m: codes.append(m.split(":")[0]), show_codes=True)
    try:
        mg.process(tp, rp,
                   [mg.pcm24(outs["lim24"]), mg.pcm16(outs["lim16"]),
                    mg.Result(outs["nolim"], "FLOAT", use_limiter=False, normalize=False),
                    mg.Result(outs["norm"], "FLOAT", use_limiter=False, normalize=True)],
                   config=cfg, preview_target=mg.pcm16(outs["pt"]), preview_result=mg.pcm16(outs["pr"]))
    finally:
        mg.log()
    for code in ("2003", "2201", "2004", "2005", "2006", "2007", "2008", "2009", "2010"):
        assert code in codes, code
    t_in, _ = audio_io.read_wav(tp)
    r_in, _ = audio_io.read_wav(rp)
    r_in = np.repeat(r_in, 2, axis=1)
    want = mo.master(t_in, r_in, mo.params(max_piece_size=1.0), True, True, True)
    got24, rate = audio_io.read_wav(outs["lim24"])
    assert rate == sr and rms_error(got24, want[0]) <= RMS_TOL
    got16, _ = audio_io.read_wav(outs["lim16"])
    assert rms_error(got16, want[0]) <= 3e-5                      # 16-bit quantisation
    assert rms_error(audio_io.read_wav(outs["nolim"])[0], want[1]) <= RMS_TOL
    assert rms_error(audio_io.read_wav(outs["norm"])[0], want[2]) <= RMS_TOL
    # previews (preview_creator.py:30-94), cut on the device: loudest 6 s window of the limited result, the
    # same frames of the clipped target, both faded over 0.5 s -- against the same cut of the oracle's arrays
    size, step, fade = 6 * sr, int(1.5 * sr), int(0.5 * sr)
    begin, want_t, want_r = expected_previews(t_in, want[0], size, step, fade, cfg.threshold)
    got_pr, got_pt = audio_io.read_wav(outs["pr"])[0], audio_io.read_wav(outs["pt"])[0]
    assert got_pr.shape == (size, 2) and got_pt.shape == (size, 2)
    assert rms_error(got_pr, want_r) <= 3e-5 and rms_error(got_pt, want_t) <= 3e-5       # 16-bit files
    assert np.abs(got_pr - want_r).max() <= 1.5 / 32768 and np.abs(got_pt - want_t).max() <= 1.5 / 32768


def expected_previews(target, result, size, step, fade, threshold):
    """preview_creator.py:30-94 + dsp.py:128-152 restated on host arrays (float64)."""
    result = np.asarray(result, dtype=np.float64)
    n = result.shape[0]
    if size > n:
        starts, size = np.array([0]), n
    else:
        starts = np.arange((n - size) // step + 1) * step
    rms = [np.sqrt(np.mean(result[s:s + size] ** 2)) for s in starts]
    begin = int(starts[int(np.argmax(rms))])
    pieces = []
    for x in (np.clip(np.asarray(target, dtype=np.float64), -threshold, threshold), result):
        piece = x[begin:begin + size].copy()
        if size != n:
            ramp = np.linspace(0, 1, fade)
            piece[:fade] *= ramp[:, None]
            piece[size - fade:] *= ramp[::-1, None]
        pieces.append(piece)
    return begin, pieces[0], pieces[1]


def test_previews_are_cut_on_the_device(tmp_path):
    """``stages.main(..., preview=PreviewRequest)``: window energies by mgx_window_energy, cut + clip + fade by
    mgx_preview_cut, float and integer pieces; the window is the one the reference's argmax picks on the
    oracle's result, the pieces equal the oracle's cut.  Also the degenerate cases of dsp.py:131-132 (a track
    shorter than the window: whole track, no fades) and fades of half a window (the ramps meet)."""
    import matchering_amd as mg
    from matchering_amd import stages
    from matchering_amd.preview import PreviewRequest
    from matchering_amd.synth import make_pair

    sr = 22050
    t, r = make_pair(16.3, sr, pair=21, reference_seconds=7.0)
    t[int(6.2 * sr):int(11.9 * sr)] *= 1.7                              # make one stretch clearly the loudest
    marker = mg.Result(str(tmp_path / "x.wav"), "FLOAT")
    for kw, encodings in ((dict(preview_size=6, preview_analysis_step=1.7, preview_fade_size=0.4), (None, None)),
                          (dict(preview_size=6, preview_analysis_step=1.7, preview_fade_size=0.4), ("PCM_16", "PCM_24")),
                          (dict(preview_size=5.5, preview_analysis_step=1.1, preview_fade_size=20, preview_fade_coefficient=2), (None, None)),
                          (dict(preview_size=30, preview_analysis_step=5), (None, None))):
        cfg = mg.Config(internal_sample_rate=sr, fft_size=1024, max_piece_size=1.5, **kw)
        req = PreviewRequest(cfg, marker, marker, encodings)
        got = stages.main(t, r, cfg, need_default=True, preview=req)[0]
        want = mo.master(t, r, mo.params(internal_sample_rate=sr, fft_size=1024, max_piece_size=1.5), True, False, False)[0]
        size, step = int(cfg.preview_size), int(cfg.preview_analysis_step)
        fade = int(min(cfg.preview_fade_size, min(size, t.shape[0]) // cfg.preview_fade_coefficient))
        begin, want_t, want_r = expected_previews(t, want, size, step, fade, cfg.threshold)
        assert req.begin == begin, (kw, req.begin, begin)
        assert req.frames == want_r.shape[0] and (req.frames == t.shape[0]) == (size > t.shape[0])
        pt, pr = req.target_piece, req.result_piece
        if encodings[0]:
            from matchering_amd import audio_io

            assert pt.dtype == np.int16 and pr.dtype == np.uint8
            pt, pr = audio_io.pcm_to_float(pt, np.float64), audio_io.pcm_to_float(pr, np.float64)
            assert rms_error(pt, want_t) <= 3e-5 and rms_error(pr, want_r) <= RMS_TOL
        else:
            assert pt.dtype == np.float32 and rms_error(pt, want_t) <= 1e-7       # the target's own frames, clipped and faded
            assert rms_error(pr, want_r) <= RMS_TOL
        # the result piece is the GPU's own result, cut and faded exactly
        _, _, own = expected_previews(t, got, size, step, fade, cfg.threshold)
        if not encodings[0]:
            assert np.abs(pr - own).max() <= 1e-7


def test_rccl_single_rank_collectives():
    """The FIR exchange of the multi-GPU path (mgx_comm_*, RCCL) on a one-rank communicator: the
    calls, the stream ordering and the payload (the FIR designed by the last mgx_master) are the ones
    bench.py uses with N ranks; only the number of peers differs."""
    import ctypes

    from matchering_amd._native import check, library
    from matchering_amd.device import default_device
    import matchering_amd as mg

    lib = library()
    dev = default_device()
    t, r = build_inputs(CASES["quiet_reference"])
    cfg = make_config(CASES["quiet_reference"]["config"])
    td, rd = dev.upload(t), dev.upload(r)
    out = dev.alloc(t.shape[0] * 8)
    dev.master(td, t.shape[0], rd, r.shape[0], cfg.to_native(), result=None, result_no_limiter=out)
    uid = ctypes.create_string_buffer(128)
    check(lib.mgx_comm_unique_id(uid))
    check(lib.mgx_comm_init(dev.handle, uid, 0, 1))
    try:
        taps_dev, taps = ctypes.c_void_p(), ctypes.c_int32()
        check(lib.mgx_last_fir(dev.handle, ctypes.byref(taps_dev), ctypes.byref(taps)))
        assert taps.value == cfg.fft_size
        count = 2 * taps.value
        table = dev.alloc(count * 4)
        check(lib.mgx_comm_allgather_f32(dev.handle, taps_dev, ctypes.c_void_p(table.ptr), count))
        check(lib.mgx_comm_broadcast_f32(dev.handle, ctypes.c_void_p(table.ptr), count, 0))
        dev.synchronize()
        own = dev.download(int(taps_dev.value), (2, taps.value))
        got = dev.download(table, (2, taps.value))
        assert np.array_equal(own, got) and np.all(np.isfinite(own)) and np.abs(own).max() > 0
    finally:
        check(lib.mgx_comm_destroy(dev.handle))


@pytest.mark.parametrize("steps", [0, 1, 7, 16, 17, 40])
def test_correction_step_counts(steps):
    """`rms_correction_steps` other than the default 4 (stages.py:149-168): none at all (the peak and
    early-out scalars then come from their own kernel), a single round (first and last at once), many -- and
    more than the 16 coefficients the report keeps (defaults.py:118-120 allows any number)."""
    import matchering_amd as mg
    from matchering_amd import stages
    from matchering_amd.synth import make_pair

    t, r = make_pair(9.0, 44100, pair=4, reference_seconds=7.0)
    kw = dict(rms_correction_steps=steps, max_piece_size=2.0)
    got = stages.main(t, r, mg.Config(**kw), need_default=True, need_no_limiter=True, need_no_limiter_normalized=True)
    want = mo.master(t, r, mo.params(**kw), True, True, True)
    for mine, ref in zip(got, want):
        assert rms_error(mine, ref) <= RMS_TOL


@pytest.mark.parametrize("case", ["shortest target", "shortest reference", "silent target", "hot target"])
def test_degenerate_inputs(case):
    """Inputs at the edges of what core.py:69-74 lets through: a track of fft_size + 1 frames (one analysis
    segment, one block pair), digital silence (every level clamps to min_value), a target far above
    full scale."""
    import matchering_amd as mg
    from matchering_amd import stages
    from matchering_amd.synth import make_pair

    t, r = make_pair(3.0, 44100, pair=2)
    if case == "shortest target":
        t = np.ascontiguousarray(t[:4097])
    elif case == "shortest reference":
        r = np.ascontiguousarray(r[:4097])
    elif case == "silent target":
        t = np.zeros_like(t)
    else:
        t = t * np.float32(8.0)
    got = stages.main(t, r, mg.Config(), need_default=True, need_no_limiter=True, need_no_limiter_normalized=True)
    want = mo.master(t, r, mo.params(), True, True, True)
    for mine, ref in zip(got, want):
        assert np.all(np.isfinite(mine))
        assert rms_error(mine, ref) <= RMS_TOL


@pytest.mark.parametrize("rate", [8000, 22050, 48000, 88200, 192000])
def test_other_sample_rates(rate):
    """Window lengths, filter poles and piece sizes all follow `internal_sample_rate` (hyrax.py:35-75,
    defaults.py:109); the limiter's chunk geometry is derived from them."""
    import matchering_amd as mg
    from matchering_amd import stages
    from matchering_amd._native import MgxError
    from matchering_amd.synth import make_pair

    seconds = 4.0 if rate <= 48000 else 2.0
    t, r = make_pair(seconds, rate, pair=6, reference_seconds=seconds * 0.8)
    kw = dict(internal_sample_rate=rate, max_piece_size=1.0)
    want = mo.master(t, r, mo.params(**kw), True, True, False)
    try:
        got = stages.main(t, r, mg.Config(**kw), need_default=True, need_no_limiter=True)
    except MgxError as exc:                        # a documented limit must say so, not miscompute
        assert exc.code == -4, exc
        pytest.skip(f"unsupported at {rate} Hz: {exc}")
    for mine, ref in zip(got[:2], want[:2]):
        assert rms_error(mine, ref) <= RMS_TOL


@pytest.mark.parametrize("fft_size", [8, 16, 32, 64, 128, 256, 512, 1024, 2048, 8192])
def test_other_fft_sizes(fft_size):
    """Every transform plan end to end: analysis segments of fft_size frames and convolution blocks of
    twice that (fft2.h plans 6..14); 8, 16 and 32 -- the smallest sizes the reference itself runs -- take the
    register transforms and the time-domain filter of small_fft_kernels.h."""
    import matchering_amd as mg
    from matchering_amd import stages
    from matchering_amd.synth import make_pair

    t, r = make_pair(3.0, 44100, pair=8, reference_seconds=2.5)
    kw = dict(fft_size=fft_size, max_piece_size=1.0)
    got = stages.main(t, r, mg.Config(**kw), need_default=True, need_no_limiter=True)
    want = mo.master(t, r, mo.params(**kw), True, True, False)
    for mine, ref in zip(got[:2], want[:2]):
        assert rms_error(mine, ref) <= RMS_TOL


def test_pinned_and_pageable_transfers_agree():
    """Device.upload moves pinned arrays by DMA where they are and stages everything else through pinned
    blocks in chunks; Device.download returns pinned arrays.  Same bytes either way, odd sizes included."""
    from matchering_amd.device import default_device, pinned

    dev = default_device()
    rng = np.random.RandomState(5)
    for n in (1, 1023, (8 << 20) // 8 + 7, 3 * (8 << 20) // 8 - 1):
        x = rng.randn(n, 2).astype(np.float32)
        px = pinned.empty(x.shape)
        px[...] = x
        assert pinned.holds(px) and not pinned.holds(x)
        with dev.lock:
            a, b = dev.upload(x), dev.upload(px)
            ya, yb = dev.download(a, x.shape, wait=False), dev.download(b, x.shape, wait=False)
            dev.synchronize()
            a.release()
            b.release()
        assert pinned.holds(ya) and np.array_equal(ya, x) and np.array_equal(yb, x)


def test_threads_sharing_the_default_device_do_not_interleave():
    """include/mgx.h: calls on one handle are serialised by the caller.  The process-wide default Device is
    what every stages.main call without an explicit device uses, so two threads of an application share
    it: Device.lock has to keep their upload -> kernels -> download sequences apart (ctypes drops the GIL
    while a call blocks).  Four threads, different pairs and shapes, many rounds; every result must be
    bit-identical to the one computed alone."""
    import threading

    import matchering_amd as mg
    from matchering_amd import stages
    from matchering_amd.synth import make_pair

    cfg = mg.Config(max_piece_size=2.0)
    pairs = [make_pair(3.0 + 0.7 * i, 44100, pair=i, reference_seconds=2.5 + 0.4 * i) for i in range(4)]
    alone = [stages.main(t, r, cfg, need_default=True, need_no_limiter=True) for t, r in pairs]
    failures = []

    def worker(i):
        try:
            for _ in range(6):
                got = stages.main(*pairs[i], cfg, need_default=True, need_no_limiter=True)
                for a, b in zip(got[:2], alone[i][:2]):
                    if not np.array_equal(a, b):
                        failures.append((i, float(np.abs(a - b).max())))
                        return
        except Exception as exc:        # noqa: BLE001 -- reported below
            failures.append((i, repr(exc)))

    threads = [threading.Thread(target=worker, args=(i,)) for i in range(4)]
    for th in threads:
        th.start()
    for th in threads:
        th.join()
    assert not failures, failures


@pytest.mark.parametrize("it", [1, 3])
def test_master_with_lowess_robustness_iterations(it):
    """Config.lowess_it > 0 (defaults.py:76, dsp.py:103-106): LOWESS is then not linear in the curve, so the
    design runs the chain on the curve itself (k_fir_direct_a, k_fir_lowess_robust, k_fir_b) instead of one
    operator product.  Against the oracle, whose LOWESS is pinned to the compiled statsmodels for it = 1..3;
    then the same Config with it = 0 (its operator is built on first need) must still agree."""
    import matchering_amd as mg
    from matchering_amd import stages
    from matchering_amd.synth import make_pair

    target, reference = make_pair(10.0, 44100, pair=4, reference_seconds=8.0)
    for passes in (it, 0):
        cfg = mg.Config(max_piece_size=2.0, fft_size=2048, lowess_it=passes)
        got = stages.main(target, reference, cfg, need_default=True, need_no_limiter=True)
        want = mo.master(target, reference, mo.params(max_piece_size=2.0, fft_size=2048, lowess_it=passes), True, True, False)
        for mine, ref in zip(got[:2], want[:2]):
            assert rms_error(mine, ref) <= RMS_TOL, (passes, rms_error(mine, ref))


def test_pcm_decode_and_encode_on_the_device_match_the_host_codec():
    """mgx_pcm_decode / mgx_pcm_encode (loader.py:35, saver.py:27-33 on the GPU) against audio_io's numpy
    codec, bit for bit: 16, 24 (packed) and 32-bit samples, lengths that leave every kind of tail, values
    beyond full scale (clipped, not wrapped) and exactly on rounding ties."""
    from matchering_amd import audio_io
    from matchering_amd.device import default_device

    dev = default_device()
    rng = np.random.RandomState(8)
    for frames in (1, 2, 3, 5, 1023, 100003):
        x = (0.5 * rng.randn(frames, 2)).astype(np.float32)
        x[rng.randint(frames), 0] = 1.7
        x[rng.randint(frames), 1] = -2.5
        x[0, 0] = np.float32(0.5 / 32767.0)                       # a tie at 16 bits: rint goes to even
        for bits in (16, 24, 32):
            q = audio_io._quantise(x, bits)
            want = audio_io._pack24(q).reshape(frames, 6) if bits == 24 else q.reshape(frames, 2)
            with dev.lock:
                buf = dev.upload(x)
                got = dev.download_pcm(buf, frames, 2, bits)
                buf.release()
            assert got.dtype == want.dtype and np.array_equal(got, want), (frames, bits)
            if bits != 24:                                        # and back: what a file of these samples decodes to
                with dev.lock:
                    back = dev.upload_frames(want)
                    floats = dev.download(back, (frames, 2))
                    back.release()
                assert np.array_equal(floats, audio_io.pcm_to_float(want, np.float32))


def test_master_on_pcm_frames_equals_master_on_their_floats():
    """stages.main fed int16 frames (a PCM_16 file as it is) and asked for PCM_16 / PCM_24 / float renderings:
    bit-identical to the float path with the host codec around it."""
    import matchering_amd as mg
    from matchering_amd import audio_io, stages
    from matchering_amd.synth import make_pair

    t, r = make_pair(6.0, 44100, pair=6, reference_seconds=5.0)
    ti = audio_io._quantise(0.6 * t, 16).reshape(t.shape)
    ri = audio_io._quantise(0.8 * r, 16).reshape(r.shape)
    cfg = mg.Config(max_piece_size=2.0)
    plain = stages.main(audio_io.pcm_to_float(ti), audio_io.pcm_to_float(ri), cfg, True, True, True)
    coded = stages.main(ti, ri, cfg, True, True, True, encodings=("PCM_16", "PCM_24", None))
    assert coded[0].dtype == np.int16 and np.array_equal(coded[0], audio_io._quantise(plain[0], 16).reshape(-1, 2))
    assert coded[1].dtype == np.uint8 and np.array_equal(coded[1], audio_io._pack24(audio_io._quantise(plain[1], 24)).reshape(-1, 6))
    assert coded[2].dtype == np.float32 and np.array_equal(coded[2], plain[2])


def test_peak_statistics_on_the_device_match_count_max_peaks():
    """mgx_peak_count (dsp.py:49-54 on the GPU) against checker.count_max_peaks, which is checked against the
    reference's own expression on the CPU: random frames, a clipped track, a limited-looking track (many
    samples within numpy.isclose's tolerance of the peak), silence."""
    from matchering_amd import checker
    from matchering_amd.device import default_device

    dev = default_device()
    rng = np.random.RandomState(21)
    cases = []
    for frames in (1, 7, 4099, 250001):
        cases.append((0.3 * rng.randn(frames, 2)).astype(np.float32))
    clipped = np.clip(1.5 * rng.randn(100000, 2), -1, 1).astype(np.float32)
    limited = (0.7 * np.tanh(3 * rng.randn(100000, 2))).astype(np.float32)
    limited[np.abs(limited) > 0.69] = np.float32(0.69) * np.sign(limited[np.abs(limited) > 0.69])
    limited[::97] *= np.float32(1 - 4e-6)                     # inside the relative tolerance of the peak
    cases += [clipped, limited, np.zeros((1000, 2), np.float32)]
    for x in cases:
        with dev.lock:
            buf = dev.upload(x)
            got = dev.peak_count(buf, x.size)
            buf.release()
        want = checker.count_max_peaks(x)
        assert got[1] == want[1] and got[0] == float(want[0]), (x.shape, got, want)


def test_process_pcm_files_stay_integer_up_to_the_gpu(tmp_path):
    """mg.process on PCM_16 and PCM_24 WAVE files without previews: samples are mapped from the files, decoded
    on the GPU (mgx_pcm_decode), the target's peak statistics come from the GPU (mgx_peak_count: the clipped
    target below must draw the warning the host-side check gives for the decoded samples -- "clipping" at 24
    bits, "a limiter was applied" at 16, where full scale reads back as 32767/32768), and the files are
    written from integers quantised on the GPU.  Against the oracle run on the decoded inputs, to a quantisation step and a half."""
    import matchering_amd as mg
    from matchering_amd import audio_io, checker
    from matchering_amd.synth import make_pair

    sr = 44100
    t, r = make_pair(8.0, sr, pair=11, reference_seconds=6.0)
    t = np.clip(1.3 * t / np.abs(t).max(), -1, 1)            # clipped on purpose
    for subtype, step in (("PCM_16", 1.0 / (1 << 15)), ("PCM_24", 1.0 / (1 << 23))):
        tp, rp = str(tmp_path / f"t_{subtype}.wav"), str(tmp_path / f"r_{subtype}.wav")
        audio_io.write_wav(tp, t, sr, subtype)
        audio_io.write_wav(rp, 0.8 * r, sr, subtype)
        out, plain = str(tmp_path / f"o_{subtype}.wav"), str(tmp_path / f"p_{subtype}.wav")
        warnings = []
        mg.log(warning_handler=lambda text: warnings.append(str(text).split(":")[0]), show_codes=True)
        try:
            mg.process(tp, rp, [mg.Result(out, subtype), mg.Result(plain, subtype, use_limiter=False, normalize=False)],
                       config=mg.Config(max_piece_size=2.0))
        finally:
            mg.log()
        ti, _ = audio_io.read_wav(tp)
        ri, _ = audio_io.read_wav(rp)
        expected = []
        mg.log(warning_handler=lambda text: expected.append(str(text).split(":")[0]), show_codes=True)
        try:
            checker.check(ti, sr, mg.Config(max_piece_size=2.0), "target")
        finally:
            mg.log()
        assert warnings == expected != [], (warnings, expected)
        want = mo.master(ti.astype(np.float64), ri.astype(np.float64), mo.params(max_piece_size=2.0), True, True, False)
        got, _ = audio_io.read_wav(out)
        assert np.abs(got - np.clip(want[0], -1, 1)).max() <= 1.6 * step + 2e-6
        got, _ = audio_io.read_wav(plain)
        assert np.abs(got - np.clip(want[1], -1, 1)).max() <= 1.6 * step + 2e-6


def test_clock_and_memory_probes_report_finite_numbers():
    """tools/probe/libmgx_probe.so (bench.py's gpu_state; measurement aid, outside include/mgx.h): not parity, and
    not a benchmark either -- on a shared box orderings between timings can flip, so this only asks for finite
    positive numbers of a plausible order of magnitude."""
    import math
    import sys

    from conftest import ROOT

    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import mgx_probe

    one = mgx_probe.clock(0, 1, 100000)
    many = mgx_probe.clock(0, 4096, 20000)
    mem = mgx_probe.memory(0)
    assert all(math.isfinite(v) and v > 0.0 for v in one + many + mem)
    assert 100.0 < one[2] <= 4000.0 and 100.0 < many[2] <= 4000.0                 # shader MHz
    assert 10.0 < mem[0] < 1e5 and 100.0 < mem[3] < 2e4                          # ns per HBM hop, GB/s streamed
