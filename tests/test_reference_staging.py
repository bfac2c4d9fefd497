"""oracle/build_ref.py: the reference byte-compiled where it lies -- a CONTAINER-ONLY staging since round 6.

The reference tree exists in the build container only, and a Python reference may not travel to the GPU box in any form
(source or byte code): ``oracle/_ref/`` is git-ignored AND gpurun-ignored.  It serves what runs here: timing the
reference's own ``stages.main`` against the oracle on one host (tools/cpu_port_vs_reference.py ->
profiles/cpu_port_vs_reference.json, which bench.py's ``cpu_baseline`` -- kind "port" on the GPU box -- carries along).
These tests (CPU) hold the recipe: nothing but ``.pyc`` files and a manifest is staged, the staged package IS the
reference (SHA-256 of every source in the manifest, its ``stages.main`` equals the oracle to rounding), and the
directory stays out of the history and out of the snapshot.
"""
import os
import subprocess
import sys

import numpy as np
import pytest

from conftest import ROOT

sys.path.insert(0, os.path.join(ROOT, "oracle"))
import build_ref  # noqa: E402


def test_the_staging_directory_stays_out_of_the_history():
    with open(os.path.join(ROOT, ".gitignore")) as fh:
        assert "oracle/_ref/" in fh.read().split()
    with open(os.path.join(ROOT, "bench.py")) as fh:        # ... and bench.py times it only where the reference tree is
        assert "if not build_ref.reference_present():" in fh.read()
    tracked = subprocess.run(["git", "ls-files", "oracle/_ref"], cwd=ROOT, capture_output=True, text=True).stdout.strip()
    assert tracked == ""
    with open(os.path.join(ROOT, ".gpurunignore")) as fh:  # ... and out of the snapshot that goes to the GPU box
        assert "oracle/_ref/" in fh.read().split()


@pytest.mark.skipif(not (build_ref.reference_present() or build_ref.staged()), reason="no reference tree and nothing staged")
def test_only_byte_code_is_staged_and_it_is_the_reference():
    if build_ref.reference_present():
        m = build_ref.build()
        assert m is not None
        import hashlib

        for rel, digest in m["files"].items():             # the manifest describes the tree the byte code came from
            with open(os.path.join(build_ref.REFERENCE_ROOT, rel), "rb") as fh:
                assert hashlib.sha256(fh.read()).hexdigest() == digest
    assert build_ref.staged()
    for folder, _dirs, names in os.walk(build_ref.STAGED):
        for name in names:
            assert name.endswith(".pyc") or name == "MANIFEST.json", os.path.join(folder, name)   # no source file
    m = build_ref.manifest()
    assert "matchering/stages.py" in m["files"] and m["python"] == list(sys.version_info[:2])


@pytest.mark.skipif(not (build_ref.reference_present() or build_ref.staged()), reason="no reference tree and nothing staged")
def test_the_staged_reference_runs_and_the_oracle_agrees_with_it():
    child = r'''
import sys, warnings
sys.path.insert(0, {root!r}); sys.path.insert(0, {root!r} + "/oracle")
import numpy as np
import build_ref, mastering_oracle as mo
from matchering_amd.synth import make_pair
if build_ref.reference_present():
    build_ref.build()
mg = build_ref.load()
from matchering import stages
assert stages.__file__.endswith(".pyc") and "/oracle/_ref/" in stages.__file__, stages.__file__
t, r = make_pair(6.0, 44100, pair=2)
with warnings.catch_warnings():
    warnings.simplefilter("ignore")
    got = stages.main(t.astype(np.float64), r.astype(np.float64), mg.Config(), need_default=True, need_no_limiter=True,
                      need_no_limiter_normalized=True)
want = mo.master(t, r, mo.params(), True, True, True)
print("MAXDIFF", max(float(np.abs(a - b).max()) for a, b in zip(got, want)))
'''.format(root=ROOT)
    done = subprocess.run([sys.executable, "-c", child], capture_output=True, text=True, timeout=300, cwd="/tmp")
    assert done.returncode == 0, done.stderr[-2000:]
    assert float(done.stdout.split("MAXDIFF", 1)[1]) <= 1e-11
