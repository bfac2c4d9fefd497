"""Sample-rate conversion for files that are not at ``Config.internal_sample_rate`` (matchering/checker.py:30-45).

The reference calls ``resampy.resample(array, sample_rate, required_sample_rate, axis=0)`` (checker.py:22,42;
requirements.txt:4, ``resampy>=0.4.2``): band-limited interpolation with resampy's default filter ``kaiser_best``.
When resampy is installed it is used as it is; where it is not (this image), ``resample`` below is the same
algorithm restated in numpy -- the Kaiser-windowed sinc table of ``resampy.filters.sinc_window(64, 9,
kaiser(beta=14.769656459379492), rolloff=0.9475937167399596)``, output sample t at input time t / ratio, left and
right filter wings with stride int(scale * 512) through the table and linear interpolation between its entries
(``resampy.interpn.resample_f``) -- evaluated as the polyphase filter it is (all phases of the rational ratio in one
prototype, scipy's ``upfirdn``: 5 s for an 8-minute stereo file) instead of sample by sample.  It replaces
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


def _prototype(plan, phases, hop):
    """resampy's weights for all `phases` fractional positions p / phases, laid out as the prototype filter of a
    polyphase resampler: with output t at input time t hop / phases = n + p / phases, the left wing's weight of
    x[n - i] sits at tap p + i phases and the right wing's weight of x[n + k + 1] at tap p - (k + 1) phases (taps
    counted from the centre).  Wing lengths, table stride (an INTEGER, int(scale * 512): the stride through the table
    is not scale * 512) and the interpolation between table entries are resampy's, phase by phase.
    Returns (taps, index of the centre tap -- a multiple of `hop`)."""
    frac = np.arange(phases) / phases
    i = np.arange(plan.taps)
    wings = []
    for right in (False, True):
        offset, eta, count = plan.wing(frac, right)
        on = i[None, :] < count[:, None]
        at = np.where(on, offset[:, None] + i[None, :] * plan.index_step, 0)
        wings.append(np.where(on, plan.win[at] + eta[:, None] * plan.delta[at], 0.0))      # [phase, tap]
    reach = plan.taps * phases
    centre = -(-reach // hop) * hop
    proto = np.zeros(2 * centre + phases)
    p = np.arange(phases)
    proto[centre + p[:, None] + i[None, :] * phases] = wings[0]
    proto[centre + p[:, None] - (i[None, :] + 1) * phases] = wings[1]
    return proto, centre


def resample(array, sample_rate, required, block=8192, max_phases=4096):
    """``array`` (n,) or (n, channels) at ``sample_rate`` -> int(n * required / sample_rate) samples at ``required``.

    required / sample_rate = L / M in lowest terms: output t sits at input time t M / L = n + p / L, and resampy's
    weights depend on the phase p only.  Its sum over the two filter wings is therefore a polyphase filter -- zero-stuff
    by L, filter with the prototype of _prototype(), keep every M-th sample -- which scipy's ``upfirdn`` evaluates
    without the zeros; samples outside the array count as zeros in both forms (resampy cuts its wings short there).  The phase arithmetic is exact (integers), where resampy's own t * (1 / ratio) carries a
    rounding of ~1e-16 t that moves a weight by ~1e-9 at the end of an hour of audio.  Ratios with more than `max_phases`
    phases, and rates that are not integers, go through the literal per-sample form.  The sums run in float64; the
    result comes back in the input's floating type, as resampy's does (float32 in, float32 out: the same array on a
    machine with resampy and on one without; integers come back as float64)."""
    from scipy.signal import upfirdn

    given = np.asarray(array).dtype
    out_dtype = given if given.kind == "f" else np.dtype(np.float64)
    x = np.ascontiguousarray(array, dtype=np.float64)
    flat = x.reshape(x.shape[0], -1)
    plan = _Plan(sample_rate, required)
    n_orig, channels = flat.shape
    n_out = int(n_orig * plan.ratio)
    y = np.zeros((n_out, channels), dtype=np.float64)
    whole = float(required) == int(required) and float(sample_rate) == int(sample_rate)
    g = np.gcd(int(required), int(sample_rate)) if whole else 1
    phases, hop = int(required) // g, int(sample_rate) // g          # L, M
    if not whole or phases > max_phases or n_out == 0:
        for t0 in range(0, n_out, block):
            _literal(plan, flat, np.arange(t0, min(n_out, t0 + block)), y)
        return y.reshape((n_out,) + x.shape[1:]).astype(out_dtype, copy=False)
    proto, centre = _prototype(plan, phases, hop)
    full = upfirdn(proto, flat, up=phases, down=hop, axis=0)
    first = centre // hop
    got = full[first:first + n_out]
    y[:got.shape[0]] = got
    return y.reshape((n_out,) + x.shape[1:]).astype(out_dtype, copy=False)
