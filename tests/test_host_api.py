"""Host side of ``process``: WAV codec, ``Result``, ``check``,
``create_preview`` -- numpy only, no GPU.  Cross-checked against the standard library's ``wave``
module and against brute-force restatements of the reference's definitions
(matchering/results.py, loader.py, saver.py, checker.py, preview_creator.py, dsp.py:128-152)."""

import wave

import numpy as np
import pytest

import matchering_amd as mg
from matchering_amd import audio_io, checker, preview
from matchering_amd.log import Code, ModuleError


def _signal(n=5000, channels=2, seed=0):
    rng = np.random.RandomState(seed)
    return np.clip(0.4 * rng.randn(n, channels), -0.999, 0.999)


# integers are written as rint(x * (2^(b-1) - 1)) and read as v / 2^(b-1) (libsndfile's scaling):
# half a step of rounding plus |x| / 2^(b-1) of scale mismatch
@pytest.mark.parametrize("subtype,tol", [("PCM_16", 1.5 / 2 ** 15), ("PCM_24", 1.5 / 2 ** 23),
                                         ("PCM_32", 1.5 / 2 ** 31), ("FLOAT", 1e-7), ("DOUBLE", 0.0),
                                         ("PCM_U8", 1.5 / 2 ** 7)])
def test_wav_round_trip(tmp_path, subtype, tol):
    x = _signal()
    path = str(tmp_path / "a.wav")
    audio_io.write_wav(path, x, 48000, subtype)
    y, rate = audio_io.read_wav(path)
    assert rate == 48000 and y.shape == x.shape
    assert np.abs(y - x).max() <= tol


@pytest.mark.parametrize("width", [2, 3])
def test_wav_agrees_with_stdlib_wave(tmp_path, width):
    x = _signal(3000)
    path = str(tmp_path / "ours.wav")
    audio_io.write_wav(path, x, 44100, "PCM_16" if width == 2 else "PCM_24")
    with wave.open(path, "rb") as w:                     # our writer, stdlib reader
        assert (w.getnchannels(), w.getsampwidth(), w.getframerate(), w.getnframes()) == (2, width, 44100, 3000)
        raw = np.frombuffer(w.readframes(3000), dtype=np.uint8).reshape(-1, width).astype(np.int64)
    v = sum(raw[:, i] << (8 * i) for i in range(width))
    v = np.where(v >= 1 << (8 * width - 1), v - (1 << (8 * width)), v).reshape(-1, 2)
    top = (1 << (8 * width - 1)) - 1
    assert np.array_equal(v, np.rint(x * top).astype(np.int64))
    theirs = str(tmp_path / "theirs.wav")                # stdlib writer, our reader
    with wave.open(theirs, "wb") as w:
        w.setnchannels(2)
        w.setsampwidth(2)
        w.setframerate(22050)
        w.writeframes(np.rint(x * 32767).astype("<i2").tobytes())
    y, rate = audio_io.read_wav(theirs)
    assert rate == 22050 and np.abs(y - np.rint(x * 32767) / 32768).max() <= 1e-12


@pytest.mark.parametrize("subtype,tol", [("PCM_S8", 1.5 / 2 ** 7), ("PCM_16", 1.5 / 2 ** 15), ("PCM_24", 1.5 / 2 ** 23),
                                         ("PCM_32", 1.5 / 2 ** 31), ("FLOAT", 1e-7), ("DOUBLE", 0.0)])
def test_aiff_round_trip(tmp_path, subtype, tol):
    x = _signal(2777)
    path = str(tmp_path / "a.aiff")
    audio_io.write_aiff(path, x, 96000, subtype)
    y, rate = audio_io.read_aiff(path)
    assert rate == 96000 and y.shape == x.shape
    assert np.abs(y - x).max() <= tol


def test_aiff_agrees_with_stdlib_aifc(tmp_path):
    aifc = pytest.importorskip("aifc")                   # in the standard library up to Python 3.12
    x = _signal(1500)
    ours = str(tmp_path / "ours.aiff")
    audio_io.write_aiff(ours, x, 44100, "PCM_24")
    with aifc.open(ours, "rb") as f:                     # our writer, stdlib reader
        assert (f.getnchannels(), f.getsampwidth(), f.getframerate(), f.getnframes()) == (2, 3, 44100, 1500)
        raw = np.frombuffer(f.readframes(1500), dtype=np.uint8).reshape(-1, 3).astype(np.int32)
    v = (raw[:, 0] << 16) | (raw[:, 1] << 8) | raw[:, 2]
    v = np.where(v & 0x800000, v - (1 << 24), v).reshape(1500, 2)
    assert np.array_equal(v, np.rint(x * (2 ** 23 - 1)).astype(np.int64))
    theirs = str(tmp_path / "theirs.aiff")               # stdlib writer, our reader
    q = np.rint(x * 32767).astype(">i2")
    with aifc.open(theirs, "wb") as f:
        f.setnchannels(2)
        f.setsampwidth(2)
        f.setframerate(22050)
        f.writeframes(q.tobytes())
    y, rate = audio_io.read_aiff(theirs)
    assert rate == 22050 and np.array_equal(y, q.astype(np.float64) / 32768.0)


def test_load_and_save_pick_the_codec_by_content_and_extension(tmp_path):
    x = _signal(900)
    path = str(tmp_path / "result.aiff")
    audio_io.save(path, x, 48000, "FLOAT")
    y, rate = audio_io.load(path, "target", str(tmp_path))
    assert rate == 48000 and np.abs(y - x).max() <= 1e-7
    assert mg.Result(str(tmp_path / "x.aiff"), subtype="FLOAT", use_limiter=False, normalize=False).subtype == "FLOAT"


def test_load_errors_follow_the_reference_codes(tmp_path):
    bad = tmp_path / "noise.bin"
    bad.write_bytes(b"this is not audio")
    for kind, code in (("target", Code.ERROR_TARGET_LOADING), ("reference", Code.ERROR_REFERENCE_LOADING)):
        with pytest.raises(ModuleError) as e:
            mg.load(str(bad), kind, str(tmp_path))
        assert e.value.code == code
    with pytest.raises(ModuleError):
        mg.load(str(tmp_path / "missing.wav"), "target", str(tmp_path))


def test_result_validation():
    r = mg.pcm16("out.wav")
    assert (r.subtype, r.use_limiter, r.normalize) == ("PCM_16", True, True)
    assert mg.pcm24("x/y.wav").subtype == "PCM_24"
    assert mg.Result("a.wav", "FLOAT", use_limiter=False, normalize=False).use_limiter is False
    with pytest.raises(TypeError, match="XYZ format is not supported"):
        mg.Result("a.xyz", "PCM_16")
    with pytest.raises(TypeError, match="WAV format does not have VORBIS subtype"):
        mg.Result("a.wav", "VORBIS")


def test_check_channels_length_and_rate():
    cfg = mg.Config()
    events = []
    mg.log(warning_handler=lambda m: events.append(("w", m)), info_handler=lambda m: events.append(("i", m)),
           show_codes=True)
    try:
        mono = _signal(50000, channels=1)
        out, rate = mg.check(mono, 44100, cfg, "reference")
        assert out.shape == (50000, 2) and np.array_equal(out[:, 0], out[:, 1]) and rate == 44100
        assert any(m.startswith("2201") for _, m in events)
        with pytest.raises(ModuleError) as e:
            mg.check(_signal(100), 44100, cfg, "target")
        assert e.value.code == Code.ERROR_TARGET_LENGTH_IS_TOO_SMALL
        with pytest.raises(ModuleError) as e:
            mg.check(np.zeros((44100 * 15 * 60 + 1, 2), dtype=np.float32), 44100, cfg, "reference")
        assert e.value.code == Code.ERROR_REFERENCE_LENGTH_LENGTH_IS_EXCEEDED
        with pytest.raises(ModuleError) as e:
            mg.check(_signal(50000, channels=3), 44100, cfg, "target")
        assert e.value.code == Code.ERROR_TARGET_NUM_OF_CHANNELS_IS_EXCEEDED
        events.clear()
        t = np.arange(48000 * 2) / 48000.0
        tone = np.stack([np.sin(2 * np.pi * 440 * t), np.sin(2 * np.pi * 440 * t)], axis=1) * 0.5
        out, rate = mg.check(tone, 48000, cfg, "target")
        assert rate == 44100 and abs(out.shape[0] - 2 * 44100) <= 1
        assert any(m.startswith("3003") for _, m in events)
        k = np.arange(out.shape[0]) / 44100.0          # the resampled tone is still the 440 Hz tone
        mid = slice(2000, -2000)
        assert np.abs(out[mid, 0] - 0.5 * np.sin(2 * np.pi * 440 * k[mid])).max() <= 2e-3
    finally:
        mg.log()


def test_check_flags_clipping_and_limiting():
    cfg = mg.Config()
    seen = []
    mg.log(warning_handler=seen.append, show_codes=True)
    try:
        x = 0.1 * np.random.RandomState(5).randn(60000, 2)
        x[100:140, 0] = 1.0                                     # 40 samples sitting at full scale
        mg.check(x, 44100, cfg, "target")
        assert any(m.startswith("3001") for m in seen)
        seen.clear()
        y = np.clip(_signal(60000, seed=3) * 3, -0.9, 0.9)      # flat-topped well below full scale
        mg.check(y, 44100, cfg, "target")
        assert any(m.startswith("3002") for m in seen)
        seen.clear()
        mg.check(0.1 * np.random.RandomState(4).randn(60000, 2), 44100, cfg, "target")
        assert not seen
    finally:
        mg.log()
    with pytest.raises(ModuleError) as e:
        checker.check_equality(x, x.copy())
    assert e.value.code == Code.ERROR_TARGET_EQUALS_REFERENCE
    checker.check_equality(x, y)


def test_preview_picks_the_loudest_window(tmp_path):
    sr = 8000
    cfg = mg.Config(internal_sample_rate=sr, fft_size=512, max_piece_size=2.0, preview_size=6,
                    preview_analysis_step=2, preview_fade_size=1)
    rng = np.random.RandomState(2)
    n = sr * 21 + 123
    result = 0.1 * rng.randn(n, 2)
    result[sr * 9: sr * 14] *= 4.0                              # loud between 9 s and 14 s
    target = 0.5 * rng.randn(n, 2)
    size, step = sr * 6, sr * 2
    rms = [np.sqrt(np.mean(result[s:s + size] ** 2)) for s in range(0, n - size + 1, step)]
    begin = int(np.argmax(rms)) * step
    pt, pr = str(tmp_path / "t.wav"), str(tmp_path / "r.wav")
    preview.create_preview(target, result, cfg, mg.Result(pt, "DOUBLE"), mg.Result(pr, "DOUBLE"))
    got_r, _ = audio_io.read_wav(pr)
    got_t, _ = audio_io.read_wav(pt)
    assert got_r.shape == (size, 2) and got_t.shape == (size, 2)
    fade = min(int(cfg.preview_fade_size), size // cfg.preview_fade_coefficient)
    ramp = np.linspace(0, 1, fade)
    want = result[begin:begin + size].copy()
    want[:fade] *= ramp[:, None]
    want[size - fade:] *= ramp[::-1, None]
    assert np.abs(got_r - want).max() <= 1e-12
    want_t = np.clip(target, -cfg.threshold, cfg.threshold)[begin:begin + size].copy()
    want_t[:fade] *= ramp[:, None]
    want_t[size - fade:] *= ramp[::-1, None]
    assert np.abs(got_t - want_t).max() <= 1e-12
    # a result shorter than the preview window is kept whole, without fades
    short = 0.1 * rng.randn(sr * 3, 2)
    preview.create_preview(short, short, cfg, None, mg.Result(pr, "DOUBLE"))
    got, _ = audio_io.read_wav(pr)
    assert np.array_equal(got, short)


def test_process_rejects_an_empty_result_list():
    with pytest.raises(RuntimeError, match="The result list is empty"):
        mg.process("a.wav", "b.wav", [])


@pytest.mark.parametrize("script", ["basic.py", "several_results.py", "custom_logging.py", "custom_config.py",
                                    "with_previews.py", "batch_of_pairs.py"])
def test_examples_use_the_api_as_it_is(script, monkeypatch):
    """Every script under examples/ runs up to its process() / process_batch() call with arguments that
    call accepts (the calls themselves need a GPU and real files, so they are intercepted)."""
    import inspect
    import os
    import runpy

    from conftest import ROOT
    from matchering_amd import batch, core

    seen = {}

    def fake_process(*args, **kwargs):
        bound = inspect.signature(core.process).bind(*args, **kwargs)
        seen["results"] = bound.arguments["results"]
        assert all(isinstance(r, mg.Result) for r in seen["results"])
        for key in ("preview_target", "preview_result"):
            assert bound.arguments.get(key) is None or isinstance(bound.arguments[key], mg.Result)
        cfg = bound.arguments.get("config")
        assert cfg is None or isinstance(cfg.to_native().fft_size, int)

    def fake_batch(*args, **kwargs):
        bound = inspect.signature(batch.process_batch).bind(*args, **kwargs)
        seen["results"] = [r for job in bound.arguments["jobs"] for r in job["results"]]
        return list(range(len(bound.arguments["jobs"])))

    monkeypatch.setattr(mg, "process", fake_process)
    monkeypatch.setattr(mg, "process_batch", fake_batch)
    monkeypatch.setattr("sys.argv", [script])
    runpy.run_path(os.path.join(ROOT, "examples", script), run_name="__main__")
    assert seen["results"]
    mg.log()                                        # examples install handlers: put the defaults back


def test_process_file_pipeline_on_cpu(tmp_path, monkeypatch):
    """core.process end to end without a GPU: load -> check -> stages.main -> save -> previews, with
    stages.main replaced by the float64 oracle (the product itself has no CPU path).  Checks which of
    the three outputs each Result receives (core.py:99-108) and the log codes' order."""
    import mastering_oracle as mo
    from matchering_amd import core
    from matchering_amd.synth import make_pair

    rate = 44100
    t, r = make_pair(8.0, rate, pair=2, reference_seconds=7.0)
    audio_io.write_wav(str(tmp_path / "t.wav"), t, rate, "FLOAT")
    audio_io.write_wav(str(tmp_path / "r.wav"), r, rate, "PCM_24")
    calls = {}

    def oracle_main(target, reference, config, need_default=True, need_no_limiter=False,
                    need_no_limiter_normalized=False, encodings=None):
        calls["needs"] = (need_default, need_no_limiter, need_no_limiter_normalized)
        assert encodings is None                       # previews were asked for: the renderings stay float
        ocfg = mo.params(max_piece_size=config.max_piece_size / config.internal_sample_rate)
        out = mo.master(audio_io.pcm_to_float(np.asarray(target), np.float64),
                        audio_io.pcm_to_float(np.asarray(reference), np.float64), ocfg,
                        need_default, need_no_limiter, need_no_limiter_normalized)
        calls["out"] = out
        return out

    monkeypatch.setattr(core, "main", oracle_main)
    codes = []
    mg.log(info_handler=lambda text: codes.append(int(str(text).split(":")[0])), show_codes=True)
    try:
        mg.process(target=str(tmp_path / "t.wav"), reference=str(tmp_path / "r.wav"),
                   results=[mg.Result(str(tmp_path / "master.wav"), "FLOAT"),
                            mg.Result(str(tmp_path / "plain.aiff"), "FLOAT", use_limiter=False, normalize=False),
                            mg.Result(str(tmp_path / "norm.wav"), "PCM_24", use_limiter=False)],
                   config=mg.Config(max_piece_size=3, preview_size=6),
                   preview_target=mg.pcm16(str(tmp_path / "pt.wav")), preview_result=mg.pcm16(str(tmp_path / "pr.wav")))
    finally:
        mg.log()
    assert calls["needs"] == (True, True, True)
    limited, plain, normalized = calls["out"]
    got, _ = audio_io.read_wav(str(tmp_path / "master.wav"))
    assert np.abs(got - limited).max() <= 1e-6
    got, _ = audio_io.read_aiff(str(tmp_path / "plain.aiff"))
    assert np.abs(got - plain).max() <= 1e-6 and np.abs(plain).max() > 1.0        # float output may exceed 0 dBFS
    got, _ = audio_io.read_wav(str(tmp_path / "norm.wav"))
    assert np.abs(got - normalized).max() <= 2.0 / 2 ** 23
    for name in ("pt.wav", "pr.wav"):
        prev, prate = audio_io.read_wav(str(tmp_path / name))
        assert prate == rate and prev.shape == (6 * rate, 2)
    # loading, exporting, previews, completed (log/codes.py); 2004-2007 come from stages.main, replaced here
    assert codes == [2003, 2008, 2009, 2010]


def test_stages_main_host_glue_with_a_stand_in_device():
    """stages.main's host side -- argument checks, uploads, which outputs are allocated, the log lines it
    builds from the device's report, clean-up -- driven with a stand-in for the Device (the oracle does
    the arithmetic).  The real Device is exercised by the GPU tests."""
    import mastering_oracle as mo
    from matchering_amd import stages
    from matchering_amd._native import MgxReport
    from matchering_amd.synth import make_pair

    class Buf:
        def __init__(self, array=None, nbytes=0):
            self.array, self.nbytes, self.released = array, nbytes, False

        def release(self):
            self.released = True

    class StandIn:
        def __init__(self):
            import threading

            self.buffers = []
            self.lock = threading.RLock()        # stages.main serialises callers of one device

        def upload(self, array, dtype=np.float32):
            b = Buf(np.ascontiguousarray(array, dtype=dtype))
            self.buffers.append(b)
            return b

        def upload_frames(self, array):
            return self.upload(audio_io.pcm_to_float(np.asarray(array)))

        def alloc(self, nbytes):
            b = Buf(nbytes=nbytes)
            self.buffers.append(b)
            return b

        def master(self, t, n, r, nr, native, result=None, result_no_limiter=None, result_no_limiter_normalized=None,
                   fir=None):
            tr = {}
            outs = mo.master(t.array.astype(np.float64), r.array.astype(np.float64),
                             mo.params(max_piece_size=2.0), result is not None, result_no_limiter is not None,
                             result_no_limiter_normalized is not None, trace=tr)
            for buf, out in zip((result, result_no_limiter, result_no_limiter_normalized), outs):
                if buf is not None:
                    buf.array = out.astype(np.float32)
            rep = MgxReport()
            rep.final_amplitude_coefficient = tr["final_amplitude_coefficient"]
            rep.rms_coefficient = tr["rms_coefficient"]
            rep.target_divisions, rep.reference_divisions = tr["target_divisions"], tr["reference_divisions"]
            rep.target_piece, rep.reference_piece = tr["target_piece"], tr["reference_piece"]
            rep.target_loud_count, rep.reference_loud_count = len(tr["target_loud_idx"]), len(tr["reference_loud_idx"])
            for i, c in enumerate(tr["correction_coefficients"]):
                rep.correction_coefficients[i] = c
            rep.normalize_coefficient = tr["normalize_coefficient"] or 0.0
            rep.limiter_active = 1
            return rep

        def download(self, buf, shape, dtype=np.float32, wait=True):
            return buf.array.reshape(shape).astype(dtype)

        def synchronize(self):
            pass

    t, r = make_pair(6.0, 44100, pair=1, reference_gain=0.5)       # quiet reference: the amplitude branch logs too
    lines = []
    mg.log(print if False else lines.append, show_codes=True)
    dev = StandIn()
    try:
        out = stages.main(t, r, mg.Config(max_piece_size=2), need_default=True, need_no_limiter=False,
                          need_no_limiter_normalized=True, device=dev)
    finally:
        mg.log()
    assert out[0].shape == t.shape and out[1] is None and out[2].shape == t.shape
    assert all(b.released for b in dev.buffers) and len(dev.buffers) == 4
    codes = [int(str(l).split(":")[0]) for l in lines if str(l)[:4].isdigit()]
    assert codes == [2004, 2005, 2006, 2007]
    text = "\\n".join(str(l) for l in lines)
    assert "correction round 4" in text and "level match" in text and "scaled back" in text
    with pytest.raises(ValueError):
        stages.main(t[:, :1], r, mg.Config(), device=dev)


def test_pcm_files_pass_through_undecoded(tmp_path):
    """audio_io: ``load(..., pcm=True)`` hands 16- and 32-bit WAVE samples over as the file holds them (they
    are decoded on the GPU, mgx_pcm_decode), ``pcm_to_float`` is the decoding the default path applies, and
    ``save`` writes integer arrays that are already quantised for the subtype byte for byte as it writes
    the floats they came from."""
    rng = np.random.RandomState(4)
    x = np.clip(0.4 * rng.randn(5003, 2), -1.2, 1.2).astype(np.float32)
    for subtype, dtype in (("PCM_16", np.int16), ("PCM_32", np.int32), ("PCM_24", np.uint8)):
        path = str(tmp_path / f"{subtype}.wav")
        audio_io.write_wav(path, x, 44100, subtype)
        plain, rate = audio_io.load(path, "target", str(tmp_path))
        raw, _ = audio_io.load(path, "target", str(tmp_path), pcm=True)
        assert rate == 44100 and raw.dtype == dtype and audio_io.pcm_channels(raw) == 2 and raw.shape[0] == x.shape[0]
        assert np.array_equal(audio_io.pcm_to_float(raw, plain.dtype), plain)
        if subtype == "PCM_16":         # written at scale 32767, read at 32768 (libsndfile): up to 1.5 steps
            assert np.abs(plain - np.clip(x, -1, 1)).max() <= 1.6 / (1 << 15)
    # integer samples in, the same file out
    for subtype, bits in (("PCM_16", 16), ("PCM_24", 24), ("PCM_32", 32)):
        q = audio_io._quantise(x, bits)
        ints = q.reshape(x.shape) if bits != 24 else audio_io._pack24(q).reshape(x.shape[0], 6)
        a, b = str(tmp_path / f"a{bits}.wav"), str(tmp_path / f"b{bits}.wav")
        audio_io.save(a, x, 44100, subtype)
        audio_io.save(b, ints, 44100, subtype)
        assert open(a, "rb").read() == open(b, "rb").read()
    with pytest.raises(TypeError):
        audio_io.write_wav(str(tmp_path / "bad.wav"), np.zeros((4, 2), np.int16), 44100, "PCM_24")


def test_check_on_integer_pcm_matches_check_on_floats():
    """checker.check / check_equality / count_max_peaks on int16 PCM (what process now feeds them) against
    the same samples as floats: same peak, same count of samples on the peak (numpy.isclose semantics,
    dsp.py:49-54), same warnings, same verdict on equality."""
    from matchering_amd import checker
    from matchering_amd.log import ModuleError

    def reference_count(array):                     # dsp.py:49-54 verbatim semantics
        m = np.abs(array).max()
        return m, int(np.count_nonzero(np.isclose(array, m) | np.isclose(array, -m)))

    rng = np.random.RandomState(2)
    for trial in range(60):
        n = int(rng.randint(50, 4000))
        ints = (np.clip(0.6 * rng.randn(n, 2), -1, 1) * 32767).astype(np.int16)
        if trial % 3 == 0:
            ints[rng.randint(n, size=20), rng.randint(2, size=20)] = ints.max()     # a limited-looking track
        floats = ints.astype(np.float64) / 32768.0
        got, want = checker.count_max_peaks(ints), reference_count(floats)
        assert got[1] == want[1] and abs(got[0] - want[0]) <= 1e-15
        got32 = checker.count_max_peaks(floats.astype(np.float32))
        assert got32[1] == reference_count(floats.astype(np.float32))[1]
    cfg = mg.Config(fft_size=64)
    seen = []
    mg.log(warning_handler=lambda text: seen.append(("int", str(text))))
    clipped = np.full((4000, 2), 32767, np.int16)
    clipped[::2] = -32768
    out, rate = checker.check(clipped[:, :1], 44100, cfg, "target")            # mono PCM: duplicated, stays PCM
    assert out.dtype == np.int16 and out.shape == (4000, 2) and rate == 44100
    mg.log(warning_handler=lambda text: seen.append(("float", str(text))))
    checker.check(clipped[:, :1].astype(np.float64) / 32768.0, 44100, cfg, "target")
    mg.log()
    assert [t for k, t in seen if k == "int"] == [t for k, t in seen if k == "float"] != []
    a = (1000 * rng.randn(300000, 2)).astype(np.int16)
    with pytest.raises(ModuleError):
        checker.check_equality(a, a.copy())
    b = a.copy()
    b[-1, 1] += 1
    checker.check_equality(a, b)                                               # one count apart: not equal
    checker.check_equality(a, a[:-1])


def test_process_keeps_pcm_integer_end_to_end(tmp_path, monkeypatch):
    """core.process on PCM_16 files without previews: the loader's int16 frames reach stages.main as they
    are, the renderings are asked for in the Results' own integer subtype when every file made from one
    rendering agrees on it, and those integers are what lands in the files."""
    from matchering_amd import core
    from matchering_amd.synth import make_pair

    rate = 44100
    t, r = make_pair(4.0, rate, pair=5, reference_seconds=3.0)
    audio_io.write_wav(str(tmp_path / "t.wav"), 0.5 * t, rate, "PCM_16")
    audio_io.write_wav(str(tmp_path / "r.wav"), 0.5 * r, rate, "PCM_16")
    seen = {}

    def fake_main(target, reference, config, need_default=True, need_no_limiter=False,
                  need_no_limiter_normalized=False, encodings=None):
        seen["dtypes"] = (target.dtype, reference.dtype)
        seen["encodings"] = encodings
        seen["needs"] = (need_default, need_no_limiter, need_no_limiter_normalized)
        base = audio_io.pcm_to_float(target)
        outs = []
        for need, fmt, gain in zip(seen["needs"], encodings, (0.9, 1.1, 1.0)):
            if not need:
                outs.append(None)
            elif fmt is None:
                outs.append(base * gain)
            else:
                bits = int(fmt[4:])
                q = audio_io._quantise(base * gain, bits)
                outs.append(audio_io._pack24(q).reshape(-1, 6) if bits == 24 else q.reshape(base.shape))
        seen["outs"] = outs
        return tuple(outs)

    monkeypatch.setattr(core, "main", fake_main)
    mg.process(str(tmp_path / "t.wav"), str(tmp_path / "r.wav"),
               [mg.pcm16(str(tmp_path / "a.wav")), mg.pcm16(str(tmp_path / "b.wav")),
                mg.Result(str(tmp_path / "c.wav"), "PCM_24", use_limiter=False, normalize=False),
                mg.Result(str(tmp_path / "d.aiff"), "PCM_16", use_limiter=False, normalize=False),
                mg.Result(str(tmp_path / "e.wav"), "PCM_24", use_limiter=False)],
               config=mg.Config(max_piece_size=2))
    assert seen["dtypes"] == (np.int16, np.int16)
    # limited: two PCM_16 WAVE files -> PCM_16; plain: a WAVE and an AIFF file -> float; normalised: PCM_24
    assert seen["encodings"] == ("PCM_16", None, "PCM_24") and seen["needs"] == (True, True, True)
    raw, _ = audio_io.read_wav(str(tmp_path / "a.wav"), pcm=True)
    assert np.array_equal(raw, seen["outs"][0])
    assert open(str(tmp_path / "a.wav"), "rb").read() == open(str(tmp_path / "b.wav"), "rb").read()
    got, _ = audio_io.read_wav(str(tmp_path / "c.wav"))
    assert np.abs(got - np.clip(seen["outs"][1], -1, 1)).max() <= 1.6 / (1 << 23)
    got, _ = audio_io.read_aiff(str(tmp_path / "d.aiff"))
    assert np.abs(got - np.clip(seen["outs"][1], -1, 1)).max() <= 1.6 / (1 << 15)
    got, _ = audio_io.read_wav(str(tmp_path / "e.wav"))
    assert np.array_equal(got, audio_io.pcm_to_float(seen["outs"][2]))


def _raw_wave(path, code, channels, bits, block, payload, rate=44100):
    import struct

    fmt = struct.pack("<HHIIHH", code, channels, rate, rate * block, block, bits)
    body = b"fmt " + struct.pack("<I", len(fmt)) + fmt + b"data" + struct.pack("<I", len(payload)) + payload
    with open(path, "wb") as fh:
        fh.write(b"RIFF" + struct.pack("<I", 4 + len(body)) + b"WAVE" + body)


def test_wave_layouts_the_codec_does_not_decode_go_to_libsndfile_or_fail_as_unrecognised(tmp_path, monkeypatch):
    """loader.py:35: soundfile reads A-law, mu-law, ADPCM ... WAVE files.  With pcm=True (what process passes)
    they must still reach libsndfile when it is installed, and without it fail as 'Format not recognised'
    (-> ffmpeg -> ModuleError 4001), never with a stray ValueError."""
    alaw = str(tmp_path / "alaw.wav")
    _raw_wave(alaw, 6, 2, 8, 2, bytes(range(200)))                 # WAVE_FORMAT_ALAW
    lying = str(tmp_path / "lying.wav")
    _raw_wave(lying, 1, 2, 16, 3, bytes(300))                      # block size 3 for 2 x 16 bits
    monkeypatch.setattr(audio_io, "_sf", None)
    for path in (alaw, lying):
        for pcm in (False, True):
            with pytest.raises(RuntimeError, match="Format not recognised"):
                audio_io._read(path, pcm)
        with pytest.raises(ModuleError) as err:
            audio_io.load(path, "target", str(tmp_path), pcm=True)
        assert "4001" in str(err.value)

    class FakeSoundfile:
        calls = []

        @classmethod
        def read(cls, path, always_2d=True):
            cls.calls.append(path)
            return np.zeros((100, 2)), 44100

    monkeypatch.setattr(audio_io, "_sf", FakeSoundfile)
    for path in (alaw, lying):
        sound, rate = audio_io._read(path, pcm=True)
        assert sound.shape == (100, 2) and rate == 44100
    assert FakeSoundfile.calls == [alaw, lying]
    # a plain PCM_16 file still comes back undecoded, without a detour through libsndfile
    plain = str(tmp_path / "plain.wav")
    audio_io.write_wav(plain, 0.25 * np.ones((64, 2)), 44100, "PCM_16")
    sound, _ = audio_io._read(plain, pcm=True)
    assert sound.dtype == np.int16 and FakeSoundfile.calls == [alaw, lying]


def test_resampling_runs_in_float64_also_for_float32_files():
    """checker.py:42 resamples the float64 arrays soundfile returns; a FLOAT (float32) track must be
    promoted before the resampler sees it."""
    rng = np.random.RandomState(3)
    audio = (0.1 * rng.randn(48000, 2)).astype(np.float32)
    cfg = mg.Config()
    out32, rate = checker.check(audio, 48000, cfg, "reference")
    out64, _ = checker.check(audio.astype(np.float64), 48000, cfg, "reference")
    assert rate == cfg.internal_sample_rate and out32.dtype == np.float64
    assert np.array_equal(out32, out64)


def test_pinned_pool_is_capped_by_total_bytes(monkeypatch):
    """free page-locked blocks are kept per size class AND under a total, the oldest go first;
    ``trim`` gives everything back.  (A stand-in allocator behind ``library()``: no GPU needed.)"""
    import ctypes
    import gc

    from matchering_amd import device

    libc = ctypes.CDLL(None)
    libc.malloc.restype = ctypes.c_void_p
    libc.free.argtypes = [ctypes.c_void_p]
    live = set()

    class FakeLibrary:
        @staticmethod
        def mgx_host_alloc(size, out):
            address = libc.malloc(ctypes.c_size_t(int(size)))
            live.add(address)
            ctypes.cast(out, ctypes.POINTER(ctypes.c_void_p))[0] = address
            return 0

        @staticmethod
        def mgx_host_free(ptr):
            live.discard(ptr.value)
            libc.free(ptr)
            return 0

    monkeypatch.setattr(device, "library", lambda: FakeLibrary)
    pool = device._PinnedPool()
    pool.MAX_FREE_BYTES = 3 << 20
    arrays = [pool.empty((1 << 18,), np.float32) for _ in range(6)]          # six 1 MiB blocks in use
    assert len(live) == 6 and all(pool.holds(a) for a in arrays)
    del arrays
    gc.collect()
    assert pool.free_bytes == 3 << 20 and len(live) == 3                      # three kept, the three oldest returns freed
    again = pool.empty((1 << 18,), np.float32)                                # recycled, not allocated
    assert len(live) == 3 and pool.free_bytes == 2 << 20
    del again
    gc.collect()
    pool.trim()
    assert pool.free_bytes == 0 and not live and not pool.starts
