"""The host-side pieces around the hot path against the UNMODIFIED reference, live: ``checker.check``
(checker.py:90-142) and ``preview_creator.create_preview`` (preview_creator.py:30-94).  Skipped where
/root/reference does not exist (the GPU box).  Resampling is excluded: the reference delegates it to
resampy, which is not installed here (matchering_amd.checker documents its stand-in)."""

import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "oracle"))

import reference_runner as rr            # noqa: E402

import matchering_amd as mg              # noqa: E402
from matchering_amd.synth import synth   # noqa: E402

pytestmark = pytest.mark.skipif(not rr.reference_available(), reason="/root/reference is not present on this machine")


def _codes(lines):
    return [line.split(":")[0].strip() for line in lines]


def _both_checks(array, rate, name, **cfg):
    import importlib

    ref = rr.load_reference()
    ref_checker = importlib.import_module("matchering.checker")
    from matchering_amd import checker

    out = []
    for module, package in ((ref_checker, ref), (checker, mg)):
        lines = []
        package.log(warning_handler=lines.append, info_handler=lines.append, show_codes=True)
        try:
            try:
                got = module.check(np.array(array), rate, package.Config(**cfg), name)
                err = None
            except Exception as exc:                 # ModuleError of either package
                got, err = None, str(exc).split(":")[0].strip()
        finally:
            package.log()
        out.append((got, _codes(lines), err))
    return out


@pytest.mark.parametrize("case", ["stereo", "mono", "clipping", "limited", "three_channels", "too_short", "too_long"])
def test_check_matches_the_reference(case):
    rate = 8000
    cfg = dict(internal_sample_rate=rate, fft_size=256, max_length=20)
    x = 0.4 * synth(3.0, rate, 3)
    name = "target"
    if case == "mono":
        x = x[:, :1]
        name = "reference"
    elif case == "clipping":
        x = np.clip(4.0 * x, -1.0, 1.0)
    elif case == "limited":
        x = np.clip(4.0 * x, -0.9, 0.9)
    elif case == "three_channels":
        x = np.concatenate((x, x[:, :1]), axis=1)
    elif case == "too_short":
        x = x[:100]
    elif case == "too_long":
        x = np.tile(x, (8, 1))
    (want, want_codes, want_err), (got, got_codes, got_err) = _both_checks(x, rate, name, **cfg)
    assert got_err == want_err and got_codes == want_codes
    if want is not None:
        assert got[1] == want[1] and got[0].shape == want[0].shape and np.array_equal(got[0], want[0])


@pytest.mark.parametrize("seconds", [2.0, 11.0, 23.0])
def test_create_preview_matches_the_reference(seconds, monkeypatch):
    import importlib

    rate = 8000
    cfg = dict(internal_sample_rate=rate, fft_size=256, preview_size=6, preview_analysis_step=2)
    rng = np.random.RandomState(int(seconds))
    target = 1.3 * synth(seconds, rate, 5)
    result = synth(seconds, rate, 6) * (0.3 + rng.rand(1)[0] * np.hanning(int(seconds * rate))[:, None])
    ref = rr.load_reference()
    ref_preview = importlib.import_module("matchering.preview_creator")
    from matchering_amd import preview

    saved = []
    for module, package in ((ref_preview, ref), (preview, mg)):
        pieces = {}
        monkeypatch.setattr(module, "save", lambda file, array, rate_, subtype, name="": pieces.__setitem__(file, np.array(array)))
        module.create_preview(np.array(target), np.array(result), package.Config(**cfg),
                              package.pcm16("target.wav"), package.pcm16("result.wav"))
        saved.append(pieces)
    want, got = saved
    for key in ("target.wav", "result.wav"):
        assert got[key].shape == want[key].shape
        assert np.abs(got[key] - want[key]).max() <= 1e-12
