"""The algebra k_correction_round / k_correction_tail rest on (csrc/mgx_kernels.h "band"), in
numpy: for an accumulated gain g in [0.7, 1.5] the sum stages.py:149-168 needs, sum clip(g * m)^2 (dsp.py:109-110),
splits into g^2 * A + C + the band's own sum, with float32 thresholds rounded towards the inside of the band."""
import numpy as np
import pytest

G_LO, G_HI = 0.7, 1.5


def thresholds():
    never = np.float32(1.0 / G_HI)
    if float(never) > 1.0 / G_HI:
        never = np.nextafter(never, np.float32(0.0))
    always = np.float32(1.0 / G_LO)
    if not float(always) > 1.0 / G_LO:
        always = np.nextafter(always, np.float32(np.inf))
    return never, always


def split(mid):
    never, always = thresholds()
    mag = np.abs(mid)
    low, high = mag <= never, mag >= always
    a = np.sum(mid[low].astype(np.float64) ** 2)
    c = float(np.count_nonzero(high))
    return a, c, mid[~low & ~high]


@pytest.mark.parametrize("seed", range(4))
def test_closed_form_parts_plus_band_equal_the_clipped_sum(seed):
    rng = np.random.RandomState(seed)
    mid = (rng.randn(200000) * rng.choice([0.2, 0.6, 1.1, 2.5])).astype(np.float32)
    mid[:64] = [np.float32(v) for v in np.linspace(1 / G_HI - 1e-6, 1 / G_HI + 1e-6, 64)]     # around the thresholds
    mid[64:128] = [np.float32(v) for v in np.linspace(1 / G_LO - 1e-6, 1 / G_LO + 1e-6, 64)]
    a, c, band = split(mid)
    for g in (G_LO, 0.83, 1.0, 1.2345, G_HI):
        want = np.sum(np.clip(mid.astype(np.float64) * g, -1.0, 1.0) ** 2)
        got = g * g * a + c + np.sum(np.clip(band.astype(np.float64) * g, -1.0, 1.0) ** 2)
        assert abs(got - want) <= 1e-12 * want


def test_thresholds_lie_inside_the_band():
    never, always = thresholds()
    assert float(never) * G_HI <= 1.0 and float(np.nextafter(never, np.float32(1.0))) * G_HI > 1.0 - 1e-7
    assert float(always) * G_LO > 1.0
    # a sample at the thresholds themselves: never clipped at the largest gain, always clipped at the smallest
    assert abs(float(never) * G_HI) <= 1.0 and abs(float(always) * G_LO) >= 1.0
