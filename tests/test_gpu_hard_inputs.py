"""GPU parity where the decisions are.

Every other GPU test draws its material from ``matchering_amd.synth`` (low-passed noise, a tone, a 7 s
loudness cycle whose piece RMS values keep clear margins).  These do not:

* a NEAR-TIE of ``rmses >= average_rms`` (match_levels.py:62-71): one piece placed 1e-7 (relative) above
  or below the average -- the GPU's loud set must be the oracle's; and a tie at 1e-12, where float32 ``mid``
  arithmetic may legitimately flip it -- then the outputs must be the oracle's WITH that piece flipped;
* a MONO target (L == R, side identically 0, side curve = A_R / min_value), also through ``process`` with
  its INFO code 2101;
* DC offset, a hard-panned track, an impulse train, a +-0.9 square wave (nearly every sample of the
  convolved mid sits in the band (1/1.5, 1/0.7) the level correction lists sample by sample);
* a small fuzz over Config x material.
All against the float64 oracle at <= 1e-5 RMS (BASELINE.json north_star).
"""
import numpy as np
import pytest

import mastering_oracle as mo
from cases import hard_material, oracle_params
from conftest import rms_error
from matchering_amd.synth import make_pair, synth

pytestmark = pytest.mark.gpu

RMS_TOL = 1e-5


def config_of(kw):
    import matchering_amd as mg

    kw = dict(kw)
    lim = kw.pop("limiter", None)
    if lim is not None:
        kw["limiter"] = mg.LimiterConfig(**lim)
    return mg.Config(**kw)


def check_against_oracle(target, reference, cfg_kw, oracle_outs=None):
    from matchering_amd import stages

    target = np.ascontiguousarray(target, dtype=np.float32)
    reference = np.ascontiguousarray(reference, dtype=np.float32)
    got = stages.main(target, reference, config_of(cfg_kw), need_default=True, need_no_limiter=True,
                      need_no_limiter_normalized=True)
    want = oracle_outs or mo.master(target, reference, oracle_params(cfg_kw), True, True, True)
    errs = []
    for name, a, b in zip(("result", "no_limiter", "no_limiter_normalized"), got, want):
        assert np.all(np.isfinite(a)), name
        errs.append(rms_error(a, b))
    return errs, got, want


# ------------------------------------------------------------------------------------------------
# near-tie of the loud-piece selection
# ------------------------------------------------------------------------------------------------
def near_tie_target(eps, piece_index=2, sr=44100, seconds=4.0, max_piece=0.5):
    """float32 target whose piece ``piece_index`` has RMS = average_rms * (1 + eps) in float64 arithmetic on
    the float32 values (as the oracle computes it), by scaling that piece."""
    cfg = mo.params(internal_sample_rate=sr, max_piece_size=max_piece, fft_size=1024)
    x = (0.5 * synth(seconds, sr, seed=77)).astype(np.float32)
    n = x.shape[0]
    divisions, piece = mo.piece_geometry(n, cfg.max_piece_size)
    lo, hi = piece_index * piece, (piece_index + 1) * piece
    base = x[lo:hi].astype(np.float64)

    def state(scale):
        y = x.copy()
        y[lo:hi] = (base * scale).astype(np.float32)
        mid = (y[:, 0].astype(np.float64) + y[:, 1].astype(np.float64)) / 2
        sq = np.array([np.mean(mid[d * piece:(d + 1) * piece] ** 2) for d in range(divisions)])
        avg = np.sqrt(sq.mean())
        return y, np.sqrt(sq[piece_index]) / avg - 1.0

    scale = 1.0
    for _ in range(60):                               # secant on the achieved ratio; float32 rounding of the piece
        y, got = state(scale)                         # moves it by ~1e-10, far below the 1e-7 aimed at
        if abs(got - eps) <= max(abs(eps) * 0.02, 3e-11):   # (re-rounding the piece to float32 jitters it by ~1e-11)
            return y, got, divisions, piece
        # d(ratio)/d(scale) ~ (1 - 1/D) / scale for a piece near the average
        scale *= 1.0 + (eps - got) / (1.0 - 1.0 / divisions)
    raise AssertionError(f"could not place the tie: last ratio {got}")


@pytest.mark.parametrize("eps", [1e-7, -1e-7])
def test_near_tie_one_part_in_ten_million_is_decided_like_the_oracle(eps):
    from matchering_amd import kernels

    target, got_eps, divisions, piece = near_tie_target(eps)
    assert (got_eps > 0) == (eps > 0) and 0.9e-7 <= abs(got_eps) <= 1.1e-7
    cfg_kw = dict(max_piece_size=0.5, fft_size=1024)
    st = kernels.analyze(target, config_of(cfg_kw), is_reference=False)
    want = mo.analyze(target.astype(np.float64), oracle_params(cfg_kw))
    assert st.divisions == divisions and st.piece_size == piece
    assert (2 in want.loud_idx) == (eps > 0)                                  # the tie piece is in or out by design
    assert np.array_equal(np.flatnonzero(st.loud), want.loud_idx)
    # float32 (L + R) / 2 on the device, float64 sums: measured 2e-9, i.e. the 1e-7 margin is 50x the noise
    assert np.abs(st.rmses / want.rmses - 1).max() <= 1e-8
    _, reference = make_pair(4.0, 44100, pair=11)
    errs, _, _ = check_against_oracle(target, reference, cfg_kw)
    assert max(errs) <= RMS_TOL, errs


@pytest.mark.parametrize("tie_piece", [2, 5])
def test_exact_tie_follows_one_of_the_two_legitimate_decisions(tie_piece):
    """Within 3e-11 the piece's float64 RMS and the average differ by less than the float32 rounding of
    (L + R) / 2 moves them (~1e-10): either decision is a correct evaluation of match_levels.py:65 on
    float32 input.  Whatever the GPU decides, its outputs must be the oracle's for THAT decision."""
    from matchering_amd import kernels

    target, got_eps, divisions, piece = near_tie_target(0.0, piece_index=tie_piece)
    assert abs(got_eps) <= 3e-11
    cfg_kw = dict(max_piece_size=0.5, fft_size=1024)
    ocfg = oracle_params(cfg_kw)
    st = kernels.analyze(target, config_of(cfg_kw), is_reference=False)
    mine = set(np.flatnonzero(st.loud).tolist())
    plain = set(mo.analyze(target.astype(np.float64), ocfg).loud_idx.tolist())
    assert mine == plain or mine == plain ^ {tie_piece}, (sorted(mine), sorted(plain))
    _, reference = make_pair(4.0, 44100, pair=11)
    original = mo.loud_pieces
    state = {"calls": 0}

    def forced(r, avg):                       # the oracle with the GPU's decision on the target's first selection
        idx, m = original(r, avg)
        state["calls"] += 1
        if state["calls"] == 1 and set(idx.tolist()) != mine:
            idx = np.array(sorted(mine))
            m = float(np.sqrt(np.mean(np.asarray(r)[idx] ** 2)))
        return idx, m

    mo.loud_pieces = forced
    try:
        want = mo.master(target, reference, ocfg, True, True, True)
    finally:
        mo.loud_pieces = original
    errs, _, _ = check_against_oracle(target, reference, cfg_kw, oracle_outs=want)
    assert max(errs) <= RMS_TOL, errs


# ------------------------------------------------------------------------------------------------
# material
# ------------------------------------------------------------------------------------------------
MATERIAL = {
    "mono_target": ("square_mono", None),
    "dc_offset": ("dc", None),
    "hard_panned_target": ("panned_chirp_impulses", None),
    "impulse_train": ("impulses", None),
    "square_wave": ("square", None),
    "square_vs_square": ("square", "square"),
    "mono_vs_panned": ("square_mono", "panned_chirp_impulses"),
}


@pytest.mark.parametrize("name", sorted(MATERIAL))
def test_hard_material_against_the_oracle(name):
    kind_t, kind_r = MATERIAL[name]
    sr = 44100
    target = hard_material(kind_t, 3.1, sr, seed=3)
    reference = hard_material(kind_r, 2.7, sr, seed=4) if kind_r else make_pair(2.7, sr, pair=13)[1]
    cfg_kw = dict(max_piece_size=0.6, fft_size=2048)
    errs, got, want = check_against_oracle(target, reference, cfg_kw)
    assert max(errs) <= RMS_TOL, (name, errs)
    # the result is brick-walled whatever went in (hyrax.py:87,97,99)
    assert np.abs(got[0]).max() <= config_of(cfg_kw).threshold * (1 + 1e-6)
    assert np.abs(got[0] - want[0]).max() <= 1e-4                      # no isolated wild sample either


def test_square_wave_fills_the_band_lists():
    """A square wave whose convolved mid lands inside (1/1.5, 1/0.7) puts (nearly) every sample of every
    workgroup into the level correction's band lists: their capacity, not only their arithmetic."""
    sr = 44100
    n = int(6.0 * sr)
    t = np.arange(n) / sr
    x = 0.5 * np.sign(np.sin(2 * np.pi * 60.0 * t))
    target = np.stack([x, x], axis=1)
    reference = np.stack([1.8 * x[: int(5.0 * sr)], 1.8 * x[: int(5.0 * sr)]], axis=1).clip(-1, 1)
    cfg_kw = dict(max_piece_size=1.0, fft_size=1024)
    tr = {}
    want = mo.master(target.astype(np.float32), reference.astype(np.float32), oracle_params(cfg_kw), True, True, True,
                     trace=tr)
    mid = np.abs(tr["result_no_limiter"].sum(axis=1) / 2 / np.prod(tr["correction_coefficients"]))
    in_band = np.mean((mid > 1 / 1.5) & (mid < 1 / 0.7))
    assert in_band > 0.5, in_band                                      # the stimulus does what it is meant to
    errs, _, _ = check_against_oracle(target, reference, cfg_kw, oracle_outs=want)
    assert max(errs) <= RMS_TOL, errs


def test_mono_target_through_process_reports_2101(tmp_path):
    """core.py:52-74 + checker.py:96-100: a one-channel target file is doubled (dsp.py:45-46) and announced
    with INFO code 2101; the result equals the oracle's on the doubled track."""
    import matchering_amd as mg
    from matchering_amd import audio_io

    sr = 44100
    mono = hard_material("square_mono", 3.0, sr, seed=5)[:, :1]
    _, reference = make_pair(2.5, sr, pair=17)
    tp, rp, op = (str(tmp_path / f) for f in ("target.wav", "reference.wav", "result.wav"))
    audio_io.write_wav(tp, mono, sr, "FLOAT")
    audio_io.write_wav(rp, reference, sr, "FLOAT")
    seen = []
    mg.log(info_handler=seen.append, show_codes=True)
    try:
        mg.process(tp, rp, [mg.Result(op, "FLOAT")], config=mg.Config(max_piece_size=0.6, fft_size=2048))
    finally:
        mg.log()
    assert any(str(m).startswith("2101") for m in seen), seen
    got, rate = audio_io.read_wav(op)
    doubled = np.repeat(mono.astype(np.float32), 2, axis=1)
    want = mo.master(doubled, reference, oracle_params(dict(max_piece_size=0.6, fft_size=2048)), True, False, False)[0]
    assert rate == sr and rms_error(got, want) <= RMS_TOL


# ------------------------------------------------------------------------------------------------
# fuzz: Config x material, small sizes
# ------------------------------------------------------------------------------------------------
def _fuzz_cases(count=14, seed=2024):
    rng = np.random.RandomState(seed)
    kinds = ["synth", "square_mono", "dc", "panned_chirp_impulses", "impulses", "square"]
    out = []
    for i in range(count):
        sr = int(rng.choice([8000, 16000, 22050, 44100, 48000]))
        fft = int(rng.choice([256, 512, 1024, 2048, 4096]))
        seconds = float(rng.uniform(1.2, 3.5)) + fft / sr
        cfg = dict(internal_sample_rate=sr, fft_size=fft, max_piece_size=float(rng.uniform(0.3, 1.5)),
                   rms_correction_steps=int(rng.randint(0, 6)), lin_log_oversampling=int(rng.choice([1, 2, 4])),
                   threshold=float(rng.choice([(2 ** 15 - 61) / 2 ** 15, 0.9, 0.5])),
                   limiter=dict(attack=float(rng.choice([0.5, 1.0, 2.0, 4.0])), hold=float(rng.choice([0.5, 1.0, 3.0])),
                                release=float(rng.choice([300.0, 1000.0, 3000.0]))))
        out.append(dict(index=i, sr=sr, seconds=seconds, kind_t=str(rng.choice(kinds)), kind_r=str(rng.choice(kinds)),
                        target_gain=float(rng.uniform(0.05, 1.5)), reference_gain=float(rng.uniform(0.4, 5.0)),
                        cfg=cfg))
    return out


def _material(kind, seconds, sr, seed):
    return synth(seconds, sr, seed=seed) if kind == "synth" else hard_material(kind, seconds, sr, seed)


@pytest.mark.parametrize("case", _fuzz_cases(), ids=lambda c: f"fuzz{c['index']}")
def test_fuzz_config_and_material(case):
    sr = case["sr"]
    target = (case["target_gain"] * _material(case["kind_t"], case["seconds"], sr, 40 + case["index"])).astype(np.float32)
    reference = np.clip(case["reference_gain"] * _material(case["kind_r"], 0.8 * case["seconds"], sr, 80 + case["index"]),
                        -1.0, 1.0).astype(np.float32)
    errs, _, _ = check_against_oracle(target, reference, case["cfg"])
    assert max(errs) <= RMS_TOL, (case, errs)


# ------------------------------------------------------------------------------------------------
# samples that are not numbers
# ------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("where, value", [("target", np.nan), ("target", np.inf), ("reference", np.nan),
                                          ("reference", -np.inf)])
def test_non_finite_samples_fail_like_the_reference(where, value):
    """The reference does not check its input (stages.py:210-272) and does not survive it either: with one NaN or
    infinity in a track no piece compares as loud, the loud selection is empty and match_frequencies.py:42 raises
    (the oracle stops one step earlier, mastering_oracle.loud_pieces).  Here the level analysis notices the same
    thing on the device, and the call -- or, without a report, the next blocking call -- fails with
    MGX_ERR_ARGUMENT; the handle is good for the next pair."""
    from matchering_amd import stages
    from matchering_amd._native import MgxError

    target, reference = make_pair(4.0, 44100, pair=2, reference_seconds=3.5)
    bad_t, bad_r = target.copy(), reference.copy()
    (bad_t if where == "target" else bad_r)[50000, 1] = value
    with pytest.raises(Exception):
        mo.master(bad_t, bad_r, mo.params(max_piece_size=1.0), True, False, False)
    with pytest.raises(MgxError, match="not finite") as caught:
        stages.main(bad_t, bad_r, config_of(dict(max_piece_size=1.0)))
    assert caught.value.code == -1
    errs, _, _ = check_against_oracle(target, reference, dict(max_piece_size=1.0))
    assert max(errs) <= RMS_TOL
