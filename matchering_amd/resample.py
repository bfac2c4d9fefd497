"""Sample-rate conversion for files that are not at ``Config.internal_sample_rate`` (matchering/checker.py:30-45).

The reference calls ``resampy.resample(array, sample_rate, required_sample_rate, axis=0)`` (checker.py:22,42;
requirements.txt:4, ``resampy>=0.4.2``): band-limited interpolation with resampy's default filter ``kaiser_best``.
When resampy is installed it is used as it is; where it is not (this image), ``resample`` below is the same
algorithm restated in numpy -- the Kaiser-windowed sinc table of ``resampy.filters.sinc_window(64, 9,
kaiser(beta=14.769656459379492), rolloff=0.9475937167399596)``, output sample t at input time t / ratio, left and
right filter wings with stride int(scale * 512) through the table and linear interpolation between its entries
(``resampy.interpn.resample_f``) -- evaluated a block of output samples at a time instead of one.  It replaces
the polyphase resampler of scipy used until round 3, which is a different filter and agreed with resampy only
to the audible, not to the numerical.  Parity with the package itself is unpinned (it is nowhere in the image);
tests/test_resample.py holds this file against a literal restatement of the package's loops
(oracle/resampy_oracle.py) and against what a band-limited resampler does to a sine.

Host code: this is the loader's path for off-rate files, not the timed path.
"""
import numpy as np

NUM_ZEROS = 64
PRECISION = 9
BETA = 14.769656459379492
ROLLOFF = 0.9475937167399596
_TABLE = None


def kaiser_best():
    """(half window, entries per zero crossing): resampy's ``kaiser_best`` table, built once per process."""
    global _TABLE
    if _TABLE is None:
        num_bits = 2 ** PRECISION
        n = num_bits * NUM_ZEROS
        sinc_win = ROLLOFF * np.sinc(ROLLOFF * np.linspace(0, NUM_ZEROS, num=n + 1, endpoint=True))
        # right half of numpy's Kaiser window of 2n + 1 points: I0(beta sqrt(1 - (k/n)^2)) / I0(beta), k = 0 .. n
        k = np.arange(n + 1, dtype=np.float64)
        taper = np.i0(BETA * np.sqrt(np.maximum(0.0, 1.0 - (k / n) ** 2))) / np.i0(BETA)
        _TABLE = (taper * sinc_win, num_bits)
    return _TABLE


class _Plan:
    """What does not depend on the samples: the table scaled for the ratio, its differences, the stride through it."""

    def __init__(self, sample_rate, required):
        self.ratio = float(required) / float(sample_rate)
        win, self.num_table = kaiser_best()
        self.win = win * self.ratio if self.ratio < 1 else win
        self.delta = np.zeros_like(self.win)
        self.delta[:-1] = np.diff(self.win)
        self.scale = min(1.0, self.ratio)
        self.index_step = int(self.scale * self.num_table)
        self.nwin = self.win.shape[0]
        self.taps = self.nwin // self.index_step + 1      # the most a wing can hold

    def wing(self, frac, right):
        """Table offsets, interpolation factors and lengths of one wing for fractional positions `frac`."""
        frac = self.scale * frac
        if right:
            frac = self.scale - frac
        index_frac = frac * self.num_table
        offset = index_frac.astype(np.int64)
        return offset, index_frac - offset, (self.nwin - offset) // self.index_step


def _literal(plan, flat, t, out):
    """resampy's loop over the output samples `t`, a block at a time: exactly its arithmetic, ends of the array included."""
    n_orig = flat.shape[0]
    i = np.arange(plan.taps)
    time_register = t * (1.0 / plan.ratio)
    n = time_register.astype(np.int64)
    acc = np.zeros((t.size, flat.shape[1]))
    for right in (False, True):
        offset, eta, limit = plan.wing(time_register - n, right)
        count = np.minimum(n_orig - n - 1, limit) if right else np.minimum(n + 1, limit)
        on = i[None, :] < count[:, None]
        at = np.where(on, offset[:, None] + i[None, :] * plan.index_step, 0)
        weight = np.where(on, plan.win[at] + eta[:, None] * plan.delta[at], 0.0)
        src = np.where(on, n[:, None] + i[None, :] + 1 if right else n[:, None] - i[None, :], 0)
        acc += np.einsum("tk,tkc->tc", weight, flat[src])
    out[t] = acc


def resample(array, sample_rate, required, block=8192, max_phases=4096):
    """``array`` (n,) or (n, channels) at ``sample_rate`` -> int(n * required / sample_rate) samples at ``required``.

    required / sample_rate = L / M in lowest terms: output t sits at input time t M / L, so the filter weights of
    outputs t and t + L are the same and their windows lie M input samples apart.  Away from the ends of the array
    (where a wing is cut short) every one of the L phases is therefore one matrix product of a strided view of the
    input with its weight vector; the phase arithmetic is exact (integers), where resampy's own t * (1 / ratio) carries
    a rounding of ~1e-16 t that moves a weight by ~1e-9 at the end of an hour of audio.  The ends, and ratios with
    more than `max_phases` phases, go through the literal per-sample form."""
    x = np.ascontiguousarray(array, dtype=np.float64)
    flat = x.reshape(x.shape[0], -1)
    plan = _Plan(sample_rate, required)
    n_orig, channels = flat.shape
    n_out = int(n_orig * plan.ratio)
    y = np.zeros((n_out, channels), dtype=np.float64)
    g = np.gcd(int(required), int(sample_rate))
    phases, hop = int(required) // g, int(sample_rate) // g          # L, M
    # outputs whose wings are complete: taps <= n(t) and n(t) + taps + 1 <= n_orig - 1, with n(t) = t M // L
    lo = -(-plan.taps * phases // hop)
    hi = min(n_out, ((n_orig - plan.taps - 2) * phases) // hop)       # (exclusive, and a little conservative)
    if phases > max_phases or float(required) != int(required) or float(sample_rate) != int(sample_rate) or hi - lo < 4 * phases:
        lo = hi = 0
    for t0 in list(range(0, lo, block)) + list(range(hi, n_out, block)):
        _literal(plan, flat, np.arange(t0, min(lo if t0 < lo else n_out, t0 + block)), y)
    step0, step1 = flat.strides
    for phase in range(phases if hi > lo else 0):
        t0 = lo + (phase - lo) % phases                               # the first interior output of this phase
        rows = (hi - 1 - t0) // phases + 1
        if rows <= 0:
            continue
        n0, rem = divmod(t0 * hop, phases)
        frac = np.array([rem / phases])
        (off_l, eta_l, cnt_l), (off_r, eta_r, cnt_r) = plan.wing(frac, False), plan.wing(frac, True)
        il, ir = np.arange(int(cnt_l[0])), np.arange(int(cnt_r[0]))
        at_l, at_r = off_l[0] + il * plan.index_step, off_r[0] + ir * plan.index_step
        left = plan.win[at_l] + eta_l[0] * plan.delta[at_l]           # weights of x[n], x[n-1], ...
        right = plan.win[at_r] + eta_r[0] * plan.delta[at_r]          # weights of x[n+1], x[n+2], ...
        weights = np.concatenate([left[::-1], right])                 # of x[n - len(left) + 1 ... n + len(right)]
        first = n0 - len(left) + 1
        view = np.lib.stride_tricks.as_strided(flat[first:], shape=(rows, len(weights), channels),
                                               strides=(hop * step0, step0, step1), writeable=False)
        y[t0:t0 + rows * phases:phases] = np.tensordot(view, weights, axes=([1], [0]))
    return y.reshape((n_out,) + x.shape[1:])
