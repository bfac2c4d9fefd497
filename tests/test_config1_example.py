"""BASELINE.json configs[0] on the MI355X path: the reference's examples/basic.py with the import changed."""
import os

import pytest

import mastering_oracle as mo
from conftest import rms_error

pytestmark = pytest.mark.gpu

RMS_TOL = 1e-5
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_config_1_examples_basic_on_a_30_second_pair(tmp_path):
    """BASELINE.json configs[0] as written, with the import changed: ``examples/basic.py`` (the reference's
    examples/basic.py:1-17) run as a script on one 30 s stereo 44.1 kHz synthetic target + reference, default Config,
    its 16- and 24-bit masters against the oracle on the decoded inputs."""
    import subprocess
    import sys

    from matchering_amd import audio_io
    from matchering_amd.synth import make_pair

    sr = 44100
    t, r = make_pair(30.0, sr, pair=3, reference_seconds=30.0)
    audio_io.write_wav(str(tmp_path / "my_song.wav"), t, sr, "PCM_16")
    audio_io.write_wav(str(tmp_path / "some_popular_song.wav"), r, sr, "PCM_16")
    env = dict(os.environ, PYTHONPATH=os.pathsep.join([ROOT, os.environ.get("PYTHONPATH", "")]))
    run = subprocess.run([sys.executable, os.path.join(ROOT, "examples", "basic.py")], cwd=str(tmp_path), env=env,
                         capture_output=True, text=True, timeout=600)
    assert run.returncode == 0, run.stdout[-2000:] + run.stderr[-2000:]
    assert run.stdout.strip(), "examples/basic.py logs every level to stdout (mg.log(print))"
    t_in, _ = audio_io.read_wav(str(tmp_path / "my_song.wav"))
    r_in, _ = audio_io.read_wav(str(tmp_path / "some_popular_song.wav"))
    want = mo.master(t_in, r_in, mo.params(), True, False, False)[0]
    got24, rate = audio_io.read_wav(str(tmp_path / "my_song_master_24bit.wav"))
    got16, _ = audio_io.read_wav(str(tmp_path / "my_song_master_16bit.wav"))
    assert rate == sr and got24.shape == want.shape
    assert rms_error(got24, want) <= RMS_TOL
    assert rms_error(got16, want) <= 3e-5                       # 16-bit quantisation
