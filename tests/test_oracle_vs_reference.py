"""The oracle against the UNMODIFIED reference, live.

``tests/test_oracle_golden.py`` pins the oracle through fixtures frozen once; this test re-runs
``/root/reference/matchering`` itself (I/O imports stubbed, oracle/reference_runner.py) on fresh
inputs each time it is collected on the build container and is skipped wherever the reference tree
does not exist (the GPU box).  Only ``stages.main`` and its helpers are exercised: the hot path.
"""

import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "oracle"))

import mastering_oracle as mo            # noqa: E402
import reference_runner as rr            # noqa: E402
from matchering_amd.synth import make_pair   # noqa: E402

pytestmark = pytest.mark.skipif(not rr.reference_available(), reason="/root/reference is not present on this machine")

CASES = {
    "cd_rate": dict(pair=dict(seconds=2.4, sample_rate=44100, pair=7, reference_seconds=2.0),
                    config=dict(max_piece_size=0.5)),
    "hot_two_rounds": dict(pair=dict(seconds=3.0, sample_rate=22050, pair=8, reference_gain=6.0),
                           config=dict(internal_sample_rate=22050, fft_size=1024, max_piece_size=0.7,
                                       rms_correction_steps=2)),
    "quiet_reference_custom_limiter": dict(
        pair=dict(seconds=1.6, sample_rate=48000, pair=9, reference_gain=0.5),
        config=dict(internal_sample_rate=48000, fft_size=2048, max_piece_size=0.4,
                    limiter=dict(attack=2.0, hold=1.5, release=800.0))),
}


@pytest.mark.parametrize("name", sorted(CASES))
def test_oracle_reproduces_the_reference(name):
    case = CASES[name]
    target, reference = make_pair(**case["pair"])
    outs_ref, inter = rr.run_reference(target, reference, case["config"])
    kw = dict(case["config"])
    kw.update(kw.pop("limiter", {}))
    trace = {}
    outs = mo.master(target, reference, mo.params(**kw), True, True, True, trace=trace)
    for mine, want in zip(outs, outs_ref):
        assert np.abs(mine - np.ascontiguousarray(want)).max() <= 1e-11
    assert abs(trace["rms_coefficient"] / inter["rms_coefficient"] - 1) <= 1e-12
    assert abs(trace["final_amplitude_coefficient"] / inter["final_amplitude_coefficient"] - 1) <= 1e-12
    assert np.abs(np.asarray(trace["correction_coefficients"]) / inter["correction_coefficients"] - 1).max() <= 1e-11
    assert np.abs(trace["fir_mid"] - inter["fir_mid"]).max() <= 1e-12 * np.abs(inter["fir_mid"]).max()
    assert np.abs(trace["fir_side"] - inter["fir_side"]).max() <= 1e-12 * np.abs(inter["fir_side"]).max()


# ---- the edges of Config the reference itself supports ------------------------------
@pytest.mark.parametrize("fft_size", [8, 16, 32])
def test_smallest_fft_sizes_the_reference_runs(fft_size):
    """defaults.py:110-112 asks only for a power of two above 1.  8, 16 and 32 run in the reference, so they run here
    (small_fft_kernels.h); the oracle is pinned to the reference there too."""
    target, reference = make_pair(3.0, 8000, pair=1)
    cfg = dict(internal_sample_rate=8000, fft_size=fft_size, max_piece_size=1.0)
    outs_ref, _ = rr.run_reference(target, reference, cfg, capture=False)
    outs = mo.master(target, reference, mo.params(**cfg), True, True, True)
    for mine, want in zip(outs, outs_ref):
        assert np.abs(mine - np.ascontiguousarray(want)).max() <= 1e-11


@pytest.mark.parametrize("fft_size", [32768, 65536])
def test_largest_fft_sizes_on_the_inputs_of_the_gpu_test(fft_size):
    """tests/test_gpu_parity.py::test_master_fft_size_32768_and_65536 compares the device with the oracle on exactly this
    pair and configuration (192 kHz, 2 s): here the oracle is held against the unmodified reference on them -- the three
    outputs and the FIR pair (65536 is the largest fft_size libmgx runs; the reference takes any power of two)."""
    sr = 192000
    target, reference = make_pair(2.0, sr, pair=14, reference_seconds=1.7)
    cfg = dict(internal_sample_rate=sr, fft_size=fft_size, max_piece_size=0.6)
    outs_ref, inter = rr.run_reference(target, reference, cfg)
    trace = {}
    outs = mo.master(target, reference, mo.params(**cfg), True, True, True, trace=trace)
    for mine, want in zip(outs, outs_ref):
        assert np.abs(mine - np.ascontiguousarray(want)).max() <= 1e-11
    assert np.abs(trace["fir_mid"] - inter["fir_mid"]).max() <= 1e-12 * np.abs(inter["fir_mid"]).max()
    assert np.abs(trace["fir_side"] - inter["fir_side"]).max() <= 1e-12 * np.abs(inter["fir_side"]).max()


@pytest.mark.parametrize("cfg", [dict(fft_size=2), dict(fft_size=4), dict(limiter=dict(hold=0.25)),
                                 dict(limiter=dict(hold=0.05))])
def test_what_the_reference_itself_cannot_run(cfg):
    """Values that pass the asserts of defaults.py and then fail INSIDE the reference: fft_size 2 and 4 (the cubic
    interp1d of match_frequencies.py:45-58 has too few points) and limiter hold times below three samples
    (hyrax.py:35-40: the sliding window is empty).  libmgx refuses them with MGX_ERR_ARGUMENT; they are not
    gaps against the reference (tests/test_gpu_parity.py::test_fails_loudly_on_unsupported)."""
    target, reference = make_pair(3.0, 8000, pair=1, reference_gain=3.0)
    kw = dict(internal_sample_rate=8000, fft_size=256, max_piece_size=1.0)
    kw.update(cfg)
    with pytest.raises(Exception):
        rr.run_reference(target, reference, kw, capture=False)


def test_more_than_sixteen_correction_steps():
    target, reference = make_pair(3.0, 8000, pair=2)
    cfg = dict(internal_sample_rate=8000, fft_size=256, max_piece_size=1.0, rms_correction_steps=20)
    outs_ref, inter = rr.run_reference(target, reference, cfg)
    trace = {}
    outs = mo.master(target, reference, mo.params(**cfg), True, True, True, trace=trace)
    assert len(inter["correction_coefficients"]) == 20
    assert np.abs(np.asarray(trace["correction_coefficients"]) / inter["correction_coefficients"] - 1).max() <= 1e-11
    for mine, want in zip(outs, outs_ref):
        assert np.abs(mine - np.ascontiguousarray(want)).max() <= 1e-11
