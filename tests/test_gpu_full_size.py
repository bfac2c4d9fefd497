"""BASELINE.json's full sizes on the GPU (configs #2/#3: one 8-minute stereo 44.1 kHz pair; config #4's
four-minute pairs): the HIP path against the float64 oracle where the oracle still finishes in
seconds, and through size-independent properties of the path (brick-wall bound, invariance to the
target's level, linearity of the convolution, limiter idempotence) where it does not need one.
Tolerance: <= 1e-5 RMS on float32 against float64 (BASELINE.json north_star), full scale = 1.0.
"""

import numpy as np
import pytest

import mastering_oracle as mo
from conftest import rms_error

pytestmark = pytest.mark.gpu

RMS_TOL = 1e-5


@pytest.fixture(scope="module")
def pair_8min():
    from matchering_amd.synth import make_pair

    return make_pair(480.0, 44100, pair=0)


@pytest.fixture(scope="module")
def mastered_8min(pair_8min):
    import matchering_amd as mg
    from matchering_amd import stages

    t, r = pair_8min
    return stages.main(t, r, mg.Config(), need_default=True, need_no_limiter=True, need_no_limiter_normalized=True)


def test_8min_pair_against_the_oracle(pair_8min, mastered_8min):
    t, r = pair_8min
    want = mo.master(t, r, mo.params(), True, True, True)
    for mine, ref in zip(mastered_8min, want):
        assert mine.shape == (t.shape[0], 2)
        assert rms_error(mine, ref) <= RMS_TOL
        assert np.abs(mine - ref).max() <= 5e-5 * max(1.0, np.abs(ref).max())


def test_8min_brick_wall_and_normalisation(mastered_8min):
    import matchering_amd as mg

    result, plain, normalized = mastered_8min
    thr = mg.Config().threshold
    assert np.abs(plain).max() > thr                       # the workload does need its limiter
    assert np.abs(result).max() <= thr * (1 + 1e-5)        # hyrax.py:78-99 is a brick wall
    assert abs(np.abs(normalized).max() / thr - 1) <= 1e-6  # dsp.py:93-100 with normalize_clipped
    assert np.all(np.isfinite(result)) and np.all(np.isfinite(plain))


def test_8min_result_does_not_depend_on_the_target_level(pair_8min, mastered_8min):
    """stages.py:80-91 matches the target's level to the reference's before anything else, so a target
    scaled by a power of two (exact in float32) must master to the same result."""
    import matchering_amd as mg
    from matchering_amd import stages

    t, r = pair_8min
    again = stages.main(t * np.float32(0.25), r, mg.Config(), need_default=True, need_no_limiter=True)
    assert rms_error(again[0], mastered_8min[0]) <= 1e-6
    assert rms_error(again[1], mastered_8min[1]) <= 1e-6


def test_8min_limiter_is_idempotent_on_its_own_output(mastered_8min):
    """A limited track stays under the threshold, so limiting it again takes the early-out
    (hyrax.py:83-85) and returns it unchanged."""
    import matchering_amd as mg
    from matchering_amd import kernels

    out, active = kernels.limit(mastered_8min[0], mg.Config(), gain=1.0, post_gain=1.0)
    assert not active
    assert np.array_equal(out, mastered_8min[0])


def test_8min_convolution_is_linear_and_shift_invariant(pair_8min):
    from matchering_amd import kernels

    rng = np.random.RandomState(1)
    f = 4096
    x = pair_8min[0]
    h1, h2 = rng.randn(f) / 64, rng.randn(f) / 64
    ya, mid_a, _ = kernels.convolve(x, h1, h2)
    yb, _, _ = kernels.convolve(x, -2 * h1, -2 * h2)
    assert rms_error(-2 * ya, yb) <= 1e-6
    assert np.abs(mid_a - 0.5 * (ya[:, 0] + ya[:, 1])).max() <= 1e-6      # dsp.py:57-68 round trip
    # delaying the input by one overlap-save block delays the output by as much (block seams are invisible)
    shift = 4096 + 37
    xs = np.zeros_like(x)
    xs[shift:] = x[:-shift]
    ys, _, _ = kernels.convolve(xs, h1, h2)
    n = x.shape[0]                                   # (the last taps of the shifted track fall off its end)
    assert rms_error(ys[shift + f:n - f], ya[f:n - shift - f]) <= 1e-6


def test_eight_4min_pairs_in_two_lanes_match_the_oracle():
    """config #4's share of one GPU."""
    import matchering_amd as mg
    from matchering_amd import batch
    from matchering_amd.synth import make_pair

    pairs = [make_pair(240.0, 44100, pair=b) for b in range(8)]
    many = batch.master_many(pairs, mg.Config(), need_default=True, lanes=2)
    for b in range(8):                                 # every pair against the oracle (~5 s of CPU each)
        want = mo.master(pairs[b][0], pairs[b][1], mo.params(), True, False, False)[0]
        assert rms_error(many[b][0], want) <= RMS_TOL
    thr = mg.Config().threshold
    assert all(np.abs(m[0]).max() <= thr * (1 + 1e-5) for m in many)


def test_96k_16k_tap_pair_at_full_size_against_the_oracle():
    """BASELINE config #5 at its full size: a four-minute 96 kHz pair (23.04 M frames, 17 pieces) with a
    16384-tap matching FIR -- the partitioned overlap-save path and the 16384-point analysis -- and the
    limiter at 96 kHz (attack window 193 frames)."""
    import matchering_amd as mg
    from matchering_amd import stages
    from matchering_amd.synth import make_pair

    t, r = make_pair(240.0, 96000, pair=5)
    kw = dict(internal_sample_rate=96000, fft_size=16384)
    got = stages.main(t, r, mg.Config(**kw), need_default=True, need_no_limiter=True)
    want = mo.master(t, r, mo.params(**kw), True, True, False)
    for mine, ref in zip(got[:2], want[:2]):
        assert rms_error(mine, ref) <= RMS_TOL
        assert np.abs(mine - ref).max() <= 5e-5 * max(1.0, np.abs(ref).max())
    assert np.abs(got[0]).max() <= mg.Config(**kw).threshold * (1 + 1e-5)


def test_maximum_length_15min_against_the_oracle():
    """The longest track the reference accepts (``max_length`` = 15 minutes, defaults.py; checker.py:58):
    39.7 M frames, 61 pieces, 4846 block pairs."""
    import matchering_amd as mg
    from matchering_amd import stages
    from matchering_amd.synth import make_pair

    t, r = make_pair(900.0, 44100, pair=3)
    got = stages.main(t, r, mg.Config(), need_default=True, need_no_limiter=True)
    want = mo.master(t, r, mo.params(), True, True, False)
    for mine, ref in zip(got[:2], want[:2]):
        assert rms_error(mine, ref) <= RMS_TOL
    assert np.abs(got[0]).max() <= mg.Config().threshold * (1 + 1e-5)
