"""The identities k_fir_taps_sub / k_fir_taps_combine (csrc/mgx_kernels.h) are built on, restated in numpy with the
kernels' own index arithmetic and checked against numpy.fft.irfft -- what match_frequencies.py:98 calls.  The GPU
parity of the taps themselves is tests/test_gpu_parity.py (golden fir_mid / fir_side, 8192 / 16384 / 32768 taps)."""
import numpy as np
import pytest


def bit_reverse(i, bits):
    return int(format(i, f"0{bits}b")[::-1], 2) if bits else 0


def sub_transform(z_in, m_len):
    """Unnormalised inverse DFT of m_len points the way a workgroup does it: bit-reversed load, one radix-2 stage when
    log2 is odd, then passes of two fused radix-2 stages with twiddles from a quarter-wave cosine table."""
    logm = m_len.bit_length() - 1
    q = np.cos(2 * np.pi * np.arange(m_len // 4 + 1) / m_len)

    def twiddle(t):                                   # exp(+2 pi i t / M), 0 <= t < M/2
        d = t - m_len // 4
        return complex(q[t] if d <= 0 else -q[m_len // 2 - t], q[abs(d)])

    z = np.zeros(m_len, complex)
    for m in range(m_len):
        z[bit_reverse(m, logm)] = z_in[m]

    def butterfly(i0, i1, w):
        y = z[i1] * w
        z[i0], z[i1] = z[i0] + y, z[i0] - y

    h = 1
    if logm & 1:
        for b in range(m_len // 2):
            butterfly(2 * b, 2 * b + 1, 1.0)
        h = 2
    while h < m_len:
        for b in range(m_len // 4):
            r = b & (h - 1)
            j = ((b - r) << 2) + r
            ta = r * (m_len // 2 // h)
            tb = ta >> 1
            butterfly(j, j + h, twiddle(ta))
            butterfly(j + 2 * h, j + 3 * h, twiddle(ta))
            butterfly(j, j + 2 * h, twiddle(tb))
            butterfly(j + h, j + 3 * h, twiddle(tb + m_len // 4))
        h <<= 2
    return z


def taps_by_split_transform(spectrum, fft, split):
    n_all, m_len = fft // 2, fft // 2 // split
    cos_table = np.cos(2 * np.pi * np.arange(fft) / fft)
    sub = []
    for rho in range(split):
        k = split * np.arange(m_len) + rho
        a, b = spectrum[k], spectrum[n_all - k]
        c, s = cos_table[k], cos_table[(k - fft // 4) & (fft - 1)]
        sub.append(sub_transform((a + b - s * (a - b)) + 1j * (c * (a - b)), m_len))
    n = np.arange(n_all)
    z = np.zeros(n_all, complex)
    for rho in range(split):
        e = (2 * rho * n) & (fft - 1)
        z += sub[rho][n & (m_len - 1)] * (cos_table[e] + 1j * cos_table[(e - fft // 4) & (fft - 1)])
    z /= fft
    h0 = np.empty(fft)
    h0[0::2], h0[1::2] = z.real, z.imag
    return h0


@pytest.mark.parametrize("fft,split", [(64, 1), (128, 1), (512, 8), (1024, 8), (4096, 8)])
def test_split_half_length_transform_is_irfft(fft, split):
    rng = np.random.RandomState(fft)
    spectrum = rng.rand(fft // 2 + 1) * 3.0
    want = np.fft.irfft(spectrum)
    got = taps_by_split_transform(spectrum, fft, split)
    assert np.abs(got - want).max() <= 4e-16 * np.abs(want).max() * np.log2(fft)


def test_tap_order_after_the_shift():
    """k_fir_taps_combine stores z[n]'s two samples at tap i = (2n + F/2) mod F: numpy.fft.ifftshift."""
    fft = 256
    h0 = np.arange(fft, dtype=float)
    shifted = np.fft.ifftshift(h0)
    taps = np.empty(fft)
    for n in range(fft // 2):
        i = (2 * n + fft // 2) & (fft - 1)
        taps[i], taps[i + 1] = h0[2 * n], h0[2 * n + 1]
    assert np.array_equal(taps, shifted)
