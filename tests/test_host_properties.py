"""Property tests (hypothesis) of the host-side code next to the hot path: the built-in RIFF/WAVE codec
(matchering/loader.py:30-47, saver.py:27-33 stand-ins), the batch sharding rule and the Config -> C struct
mapping.  CPU only."""

import numpy as np
from hypothesis import given, settings, strategies as st

from matchering_amd import audio_io, batch

SUBTYPES = {"PCM_16": 1.5 / 2 ** 15, "PCM_24": 1.5 / 2 ** 23, "PCM_32": 1.5 / 2 ** 31, "FLOAT": 1e-7, "DOUBLE": 0.0}


@settings(max_examples=40, deadline=None)
@given(frames=st.integers(1, 3000), channels=st.integers(1, 2), rate=st.sampled_from([8000, 22050, 44100, 48000, 96000]),
       subtype=st.sampled_from(sorted(SUBTYPES)), seed=st.integers(0, 2 ** 16))
def test_wav_round_trip_any_shape(tmp_path_factory, frames, channels, rate, subtype, seed):
    rng = np.random.RandomState(seed)
    x = np.clip(0.5 * rng.randn(frames, channels), -0.999, 0.999)
    path = str(tmp_path_factory.mktemp("wav") / "x.wav")
    audio_io.write_wav(path, x, rate, subtype)
    y, got_rate = audio_io.read_wav(path)
    assert got_rate == rate and y.shape == x.shape
    assert np.abs(y - x).max() <= SUBTYPES[subtype]


@settings(max_examples=40, deadline=None)
@given(frames=st.integers(1, 2000), channels=st.integers(1, 2), subtype=st.sampled_from(["PCM_16", "PCM_24", "PCM_32"]),
       seed=st.integers(0, 2 ** 16), gain=st.floats(0.1, 1.6))
def test_integer_pcm_passes_through_the_codec_untouched(tmp_path_factory, frames, channels, subtype, seed, gain):
    """``read_wav(pcm=True)`` hands over what the file holds, ``pcm_to_float`` decodes it exactly like the default
    path, and writing those integers back gives the same bytes (what the GPU's quantiser relies on): any
    shape, values beyond full scale included (clipped on writing, never wrapped)."""
    rng = np.random.RandomState(seed)
    x = (gain * rng.randn(frames, channels)).astype(np.float32)
    folder = tmp_path_factory.mktemp("pcm")
    a, b = str(folder / "a.wav"), str(folder / "b.wav")
    audio_io.write_wav(a, x, 44100, subtype)
    raw, _ = audio_io.read_wav(a, pcm=True)
    floats, _ = audio_io.read_wav(a)
    assert raw.dtype.kind in "iu" and audio_io.pcm_channels(raw) == channels and raw.shape[0] == frames
    assert np.array_equal(audio_io.pcm_to_float(raw, floats.dtype), floats)
    assert np.abs(floats).max() <= 1.0
    audio_io.write_wav(b, np.array(raw), 44100, subtype)
    with open(a, "rb") as fa, open(b, "rb") as fb:
        assert fa.read() == fb.read()


@settings(max_examples=60, deadline=None)
@given(count=st.integers(0, 200), world=st.integers(1, 16))
def test_sharding_is_a_balanced_partition(count, world):
    items = list(range(count))
    shards = [batch.shard(items, r, world) for r in range(world)]
    assert sorted(i for s in shards for i in s) == items
    sizes = [len(s) for s in shards]
    assert max(sizes) - min(sizes) <= 1                      # near-linear batch scaling needs even shards


@settings(max_examples=30, deadline=None)
@given(rate=st.sampled_from([8000, 22050, 44100, 48000, 88200, 96000, 192000]),
       attack=st.floats(0.5, 5.0), hold=st.floats(0.5, 5.0), release=st.floats(500.0, 6000.0))
def test_config_reaches_the_c_struct_unchanged(rate, attack, hold, release):
    """`Config` / `LimiterConfig` -> `mgx_config` (include/mgx.h): every field lands in its slot."""
    import matchering_amd as mg

    lim = mg.LimiterConfig(attack=attack, hold=hold, release=release)
    cfg = mg.Config(internal_sample_rate=rate, limiter=lim)
    native = cfg.to_native()
    assert (native.attack_ms, native.hold_ms, native.release_ms) == (attack, hold, release)
    assert native.internal_sample_rate == rate and native.fft_size == cfg.fft_size
    assert native.max_piece_size == cfg.max_piece_size and native.threshold == cfg.threshold
    assert native.hold_filter_coefficient == lim.hold_filter_coefficient
    assert native.release_filter_coefficient == lim.release_filter_coefficient
