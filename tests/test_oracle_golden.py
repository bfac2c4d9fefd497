"""The oracle (oracle/mastering_oracle.py) against outputs frozen from the
UNMODIFIED reference (tests/golden/*.npz, produced by make_golden.py from
/root/reference/matchering/stages.py:210-272 with the compiled statsmodels
LOWESS).  This is what pins the oracle; everything GPU-side is then compared
with the oracle.  CPU-only, no /root/reference needed at run time."""

import hashlib

import numpy as np
import pytest

import mastering_oracle as mo
from cases import CASES, build_inputs, oracle_params

TIGHT = 1e-11   # float64 restatement vs float64 reference, absolute (full scale = 1)


def tight(name):
    """The bound of a case: 1e-11 unless the case says why it cannot be (cases.py: oracle_tolerance)."""
    return CASES[name].get("oracle_tolerance", TIGHT)


@pytest.fixture(scope="module")
def runs():
    cache = {}

    def run(name):
        if name not in cache:
            target, reference = build_inputs(CASES[name])
            trace = {}
            outs = mo.master(target, reference, oracle_params(CASES[name]["config"]),
                             True, True, True, trace=trace)
            cache[name] = (target, reference, outs, trace)
        return cache[name]

    return run


@pytest.mark.parametrize("name", sorted(CASES))
def test_inputs_are_the_frozen_ones(name, golden):
    target, reference = build_inputs(CASES[name])
    digest = hashlib.sha256(target.tobytes() + reference.tobytes()).hexdigest()
    assert digest == str(golden(name)["input_sha256"]), "synthetic generator drifted; regenerate goldens"


@pytest.mark.parametrize("name", sorted(CASES))
def test_outputs_match_reference(name, golden, runs):
    g = golden(name)
    _, _, outs, _ = runs(name)
    idx = g["sparse_index"]
    for key, mine in zip(("result", "result_no_limiter", "result_no_limiter_normalized"), outs):
        assert np.abs(mine[idx] - g[key + "_sparse"]).max() <= tight(name), key
        if key + "_f32" in g:
            assert np.abs(mine - g[key + "_f32"]).max() <= 2e-7 * max(1.0, np.abs(mine).max()), key


@pytest.mark.parametrize("name", sorted(CASES))
def test_intermediates_match_reference(name, golden, runs):
    g = golden(name)
    _, _, _, tr = runs(name)
    assert tr["target_divisions"] == int(g["target_divisions"])
    assert tr["target_piece"] == int(g["target_piece"])
    assert tr["reference_divisions"] == int(g["reference_divisions"])
    assert tr["reference_piece"] == int(g["reference_piece"])
    assert len(tr["target_loud_idx"]) == int(g["target_loud_count"])
    assert len(tr["reference_loud_idx"]) == int(g["reference_loud_count"])
    for key in ("final_amplitude_coefficient", "target_match_rms", "reference_match_rms",
                "rms_coefficient"):
        assert abs(tr[key] - float(g[key])) <= 1e-12 * max(1.0, abs(float(g[key]))), key
    assert np.abs(tr["correction_coefficients"] - g["correction_coefficients"]).max() <= 1e-12
    assert np.abs(tr["fir_mid"] - g["fir_mid"]).max() <= TIGHT
    assert np.abs(tr["fir_side"] - g["fir_side"]).max() <= TIGHT
    assert abs(tr["normalize_coefficient"] - float(g["normalize_coefficient"])) <= 1e-12


@pytest.mark.parametrize("name", sorted(CASES))
def test_limiter_envelopes_match_reference(name, golden, runs):
    g = golden(name)
    _, _, _, tr = runs(name)
    env = mo.limiter_envelopes(tr["result_no_limiter"], oracle_params(CASES[name]["config"]))
    if not bool(g["limiter_active"]):
        assert env is None          # hyrax.py:83-85 early-out
        return
    idx = g["sparse_index"]
    assert np.abs(env.slided[idx] - g["limiter_slided_sparse"]).max() <= TIGHT
    assert np.abs(env.g_att[idx] - g["limiter_gain_attack_sparse"]).max() <= TIGHT
    assert np.abs(env.g_rel[idx] - g["limiter_gain_release_sparse"]).max() <= tight(name)


def test_lowess_known_answer(golden):
    g = golden("lowess_kat")
    fit = mo.lowess_it0(g["y"], float(g["frac"]), float(g["delta"]))
    assert np.abs(fit - g["fit"]).max() <= TIGHT


def test_lowess_with_robustness_iterations_known_answer(golden):
    """Config.lowess_it > 0 (defaults.py:76): the restatement against the compiled statsmodels' outputs for
    1, 2 and 3 robustness passes (tests/golden/make_lowess_robust.py; the input carries outliers)."""
    g = golden("lowess_robust_kat")
    for it in (1, 2, 3):
        fit = mo.lowess(g["y"], float(g["frac"]), float(g["delta"]), it)
        assert np.abs(fit - g[f"fit{it}"]).max() <= TIGHT
    assert np.abs(g["fit1"] - mo.lowess_it0(g["y"], float(g["frac"]), float(g["delta"]))).max() > 1e-2


def test_limiter_never_exceeds_threshold(runs):
    # invariant from hyrax.py:87,97,99: gain <= 1/rectified
    for name in ("cd_default", "hot_lowrate"):
        _, _, outs, tr = runs(name)
        cfg = oracle_params(CASES[name]["config"])
        assert np.abs(outs[0]).max() <= cfg.threshold * tr["final_amplitude_coefficient"] * (1 + 1e-12)


def test_identity_fir_is_identity():
    # 'same' offset is (F-1)//2 (match_frequencies.py:112, scipy _centered)
    rng = np.random.RandomState(0)
    m, s = rng.randn(5000), rng.randn(5000)
    f = 256
    delta = np.zeros(f)
    delta[(f - 1) // 2] = 1.0
    y, ym = mo.convolve_same(m, delta, s, delta)
    assert np.abs(ym - m).max() <= 1e-12
    assert np.abs(y[:, 0] - (m + s)).max() <= 1e-12 and np.abs(y[:, 1] - (m - s)).max() <= 1e-12
