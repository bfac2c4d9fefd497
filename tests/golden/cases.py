"""Golden-case definitions shared by make_golden.py and the tests.

A case = recipe for the synthetic inputs + ``Config`` keyword arguments (in the
reference's constructor units: seconds for ``max_piece_size``; a nested
``limiter`` dict holds ``LimiterConfig`` keyword arguments)."""

import numpy as np

from matchering_amd.synth import make_pair

CASES = {
    # CD-rate defaults (F = 4096), several analysis pieces, limiter engaged on ~0.7 % of frames
    "cd_default": dict(
        seconds=2.0, reference_seconds=1.7, sample_rate=44100, pair=0,
        config=dict(max_piece_size=0.45)),
    # low rate, short FIR, "hot" reference: ~25 % of the frames are limited, odd length
    "hot_lowrate": dict(
        seconds=6.0125, reference_seconds=4.9, sample_rate=8000, pair=1, reference_gain=6.0,
        config=dict(internal_sample_rate=8000, fft_size=512, max_piece_size=1.3)),
    # reference below the threshold: final amplitude coefficient != 1 (match_levels.py:29-44)
    "quiet_reference": dict(
        seconds=3.0, reference_seconds=4.1, sample_rate=22050, pair=2, reference_gain=0.6,
        config=dict(internal_sample_rate=22050, fft_size=1024, max_piece_size=0.8)),
    # non-default limiter timing and pipeline knobs
    "custom_limiter": dict(
        seconds=1.5, reference_seconds=1.5, sample_rate=48000, pair=3, reference_gain=4.0,
        config=dict(internal_sample_rate=48000, fft_size=2048, max_piece_size=0.4,
                    rms_correction_steps=2, lin_log_oversampling=2,
                    limiter=dict(attack=2.5, hold=3.0, release=1000.0,
                                 hold_filter_coefficient=11.0, release_filter_coefficient=600.0))),
    # what round 2 added to the supported configurations, through the reference itself: LOWESS with two
    # robustness passes (statsmodels compiled, defaults.py:76) and second-order hold / release filters
    # (defaults.py:43-47, hyrax.py:55-72)
    # (second-order sections 1e-5 from z = 1 amplify the last bits in which two scipy versions' butter()
    # coefficients differ by ~1e5: the restatement under this numpy / scipy and the reference under the conda
    # interpreter's agree to 5e-11 instead of 1e-11)
    "robust_second_order": dict(
        oracle_tolerance=2e-10,
        seconds=3.0, reference_seconds=2.6, sample_rate=44100, pair=5, reference_gain=3.0,
        config=dict(fft_size=2048, max_piece_size=0.7, lowess_it=2,
                    limiter=dict(hold_filter_order=2, release_filter_order=2))),
    # limiter timings at both ends of what the kernels handle: a 4-sample attack and hold (frame-by-frame
    # windows), and an 8 ms attack whose halos need 1024-block chunks
    "limiter_four_samples": dict(
        seconds=2.0, reference_seconds=1.8, sample_rate=44100, pair=7, reference_gain=4.0,
        config=dict(fft_size=1024, max_piece_size=0.5, limiter=dict(attack=0.1, hold=0.1))),
    "limiter_long_attack": dict(
        seconds=2.5, reference_seconds=2.0, sample_rate=44100, pair=8, reference_gain=4.0,
        config=dict(fft_size=1024, max_piece_size=0.6, limiter=dict(attack=8.0, hold=2.0))),
    # the largest transform: fft_size 32768 at 192 kHz (analysis segments of two 16384-point transforms, the FIR
    # designed on the curve, 32 k taps in eight partitions)
    "fft_32768": dict(
        seconds=1.6, reference_seconds=1.3, sample_rate=192000, pair=6,
        config=dict(internal_sample_rate=192000, fft_size=32768, max_piece_size=0.5)),
    # reference whose peak is one short burst: after peak normalisation its RMS is low, the
    # result never reaches the threshold and the limiter early-outs (hyrax.py:83-85)
    "limiter_bypassed": dict(
        kind="burst_reference", seconds=2.5, reference_seconds=2.0, sample_rate=16000, pair=4,
        reference_gain=0.5,
        config=dict(internal_sample_rate=16000, fft_size=1024, max_piece_size=0.7)),
    # NOT from matchering_amd.synth: a mono target (L == R: the side channel is
    # identically zero, so the side matching curve is A_R / min_value ~ 1e6) made of a 110 Hz square wave with a
    # DC offset, against a hard-panned reference (R == 0) made of a chirp and an impulse train
    "square_mono_vs_panned": dict(
        kind="square_mono_vs_panned", seconds=2.2, reference_seconds=2.0, sample_rate=44100, pair=9,
        config=dict(fft_size=2048, max_piece_size=0.5)),
}


def hard_material(kind, seconds, sample_rate, seed=0):
    """Programme material that is nothing like synth(): the shapes the level decisions and the band lists of
    the level correction have to survive (square waves, impulse trains, DC, one dead channel)."""
    n = int(seconds * sample_rate)
    t = np.arange(n) / sample_rate
    rng = np.random.RandomState(1000 + seed)
    if kind == "square_mono":                 # L == R exactly
        x = 0.45 * np.sign(np.sin(2 * np.pi * 110.0 * t)) * (0.6 + 0.4 * np.sin(2 * np.pi * t / 1.3)) + 0.03
        x = x + 1e-4 * rng.randn(n)
        return np.stack([x, x], axis=1)
    if kind == "panned_chirp_impulses":       # R == 0 exactly
        f0, f1 = 50.0, 8000.0
        phase = 2 * np.pi * (f0 * t + (f1 - f0) * t * t / (2 * seconds))
        left = 0.8 * np.sin(phase) * (0.5 + 0.5 * np.sin(2 * np.pi * t / 0.9) ** 2)
        left[::997] = 0.95
        return np.stack([left, np.zeros(n)], axis=1)
    if kind == "square":                      # +-0.9 square, channels in opposite phase half of the time
        x = 0.9 * np.sign(np.sin(2 * np.pi * 82.0 * t))
        y = x * np.sign(np.sin(2 * np.pi * 0.7 * t) + 0.3)
        return np.stack([x, y], axis=1) * (0.7 + 0.3 * np.sin(2 * np.pi * t / 1.7))[:, None]
    if kind == "impulses":                    # sparse full-scale clicks over a -60 dB floor
        x = 1e-3 * rng.randn(n, 2)
        x[::613, 0] += 0.9
        x[305::613, 1] -= 0.9
        return x
    if kind == "dc":                          # music-like noise riding on a large DC offset
        x = 0.1 * rng.randn(n, 2) * (0.5 + 0.5 * np.sin(2 * np.pi * t / 1.1) ** 2)[:, None]
        return x + np.array([0.3, -0.2])
    raise ValueError(kind)


def build_inputs(case):
    sr = case["sample_rate"]
    if case.get("kind") == "square_mono_vs_panned":
        target = hard_material("square_mono", case["seconds"], sr, case["pair"])
        reference = hard_material("panned_chirp_impulses", case["reference_seconds"], sr, case["pair"])
        return target.astype(np.float32), reference.astype(np.float32)
    target, reference = make_pair(
        case["seconds"], sr, case["pair"], reference_seconds=case.get("reference_seconds"),
        reference_gain=case.get("reference_gain", 2.5))
    if case.get("kind") == "burst_reference":
        n = reference.shape[0]
        reference = reference.copy()
        reference[n // 2 : n // 2 + int(0.02 * sr)] *= 4.0
    return target, reference


def oracle_params(cfg_kwargs):
    import mastering_oracle as mo

    kw = dict(cfg_kwargs)
    kw.update(kw.pop("limiter", {}))
    return mo.params(**kw)


def sparse_index(n):
    """Frames kept in float64: both edges (filter edge effects live there) and
    every 13th frame in between."""
    edge = min(3000, n // 3)
    return np.unique(np.concatenate((np.arange(edge), np.arange(edge, n - edge, 13),
                                     np.arange(n - edge, n))))
