"""Stage the UNMODIFIED reference for the CPU baseline -- TEST / MEASUREMENT INFRASTRUCTURE, never the product.

    python oracle/build_ref.py            (called by __graft_entry__.build() when /root/reference is present)

The reference (sergree/matchering) is pure Python: there is nothing to compile with gcc, and its sources must not be
copied into this repository.  What `bench.py`'s `cpu_baseline` leg needs on the GPU box -- where /root/reference does
not exist -- is the reference's own `stages.main` (stages.py:210-272), runnable.  So this recipe does for Python what a
Makefile does for a C reference: it COMPILES the package where it lies, `/root/reference/matchering/**/*.py` ->
sourceless byte code `oracle/_ref/matchering/**/*.pyc` (py_compile, the interpreter of this image = the interpreter of
the GPU box), and writes nothing else but a manifest (file names, SHA-256 of every source, interpreter version).
`oracle/_ref/` is git-ignored (it stays out of the history) and travels to the GPU box with the snapshot, like the
built `.so` files.  No source line of the reference enters the tree.

`load()` imports the staged package with the same three stand-ins as oracle/reference_runner.py:
empty `soundfile` / `resampy` modules (file I/O only, never reached from stages.main) and `statsmodels.api`'s one call
site (dsp.py:103-106, LOWESS) served by the oracle's restatement, which tests/golden pins to the compiled statsmodels
(<= 1e-12).  Only `bench.py`'s cpu_baseline leg and tests call this.
"""
import hashlib
import importlib
import json
import os
import py_compile
import sys
import types

HERE = os.path.dirname(os.path.abspath(__file__))
REFERENCE_ROOT = "/root/reference"
STAGED = os.path.join(HERE, "_ref")
MANIFEST = os.path.join(STAGED, "MANIFEST.json")


def reference_present():
    return os.path.isdir(os.path.join(REFERENCE_ROOT, "matchering"))


def staged():
    """True when a byte-compiled reference made by THIS interpreter version is in place."""
    try:
        with open(MANIFEST) as fh:
            m = json.load(fh)
    except (OSError, ValueError):
        return False
    return (m.get("python") == list(sys.version_info[:2])
            and os.path.exists(os.path.join(STAGED, "matchering", "__init__.pyc")))


def build(force=False):
    """Byte-compile the reference package into oracle/_ref/.  Returns the manifest, or None when there is no reference
    tree here (the GPU box: the staged files that came with the snapshot are used as they are)."""
    if not reference_present():
        return None
    src_root = os.path.join(REFERENCE_ROOT, "matchering")
    files = {}
    for folder, _dirs, names in os.walk(src_root):
        for name in sorted(names):
            if name.endswith(".py"):
                path = os.path.join(folder, name)
                with open(path, "rb") as fh:
                    files[os.path.relpath(path, REFERENCE_ROOT)] = hashlib.sha256(fh.read()).hexdigest()
    manifest = {"python": list(sys.version_info[:2]), "source": "sergree/matchering, /root/reference (unmodified)",
                "files": files}
    if not force and staged():
        with open(MANIFEST) as fh:
            if json.load(fh).get("files") == files:
                return manifest
    for rel in files:
        dst = os.path.join(STAGED, rel + "c")                      # x.py -> x.pyc beside where the source would be
        os.makedirs(os.path.dirname(dst), exist_ok=True)
        py_compile.compile(os.path.join(REFERENCE_ROOT, rel), cfile=dst, dfile=rel, doraise=True, optimize=0)
    version = None
    try:
        with open(os.path.join(src_root, "__init__.py")) as fh:
            for line in fh:
                if line.startswith("__version__"):
                    version = line.split("=", 1)[1].strip().strip("\"'")
    except OSError:
        pass
    manifest["version"] = version
    with open(MANIFEST, "w") as fh:
        json.dump(manifest, fh, indent=1, sort_keys=True)
    return manifest


def manifest():
    with open(MANIFEST) as fh:
        return json.load(fh)


_installed = []


def unload():
    """Take the stubs ``load`` installed, the staged package and its path out of this process again."""
    for name in list(_installed):
        sys.modules.pop(name, None)
    del _installed[:]
    for name in [n for n in sys.modules if n == "matchering" or n.startswith("matchering.")]:
        if STAGED in (getattr(sys.modules[name], "__file__", "") or ""):
            sys.modules.pop(name, None)
    while STAGED in sys.path:
        sys.path.remove(STAGED)


def load():
    """Import the staged reference (sourceless).  Raises ImportError when it is not staged."""
    if not staged():
        raise ImportError("oracle/_ref holds no byte-compiled reference for this interpreter (python oracle/build_ref.py)")
    import numpy as np

    if HERE not in sys.path:
        sys.path.insert(0, HERE)
    from mastering_oracle import lowess_it0

    def lowess(endog, exog, frac, it, delta):                    # statsmodels.api.nonparametric.lowess (dsp.py:103-106)
        if it != 0:
            raise NotImplementedError("the LOWESS stand-in of the CPU baseline covers lowess_it = 0 (the default)")
        return np.stack((exog, lowess_it0(endog, frac, delta)), axis=1)

    def _not_here(*_a, **_k):
        raise RuntimeError("soundfile / resampy are stubs of oracle/build_ref.py: stages.main never reaches them")

    # (the reference imports both at module level -- checker.py:22 `from resampy import resample` -- so the stubs must
    # carry the names; whoever loads this into a process that also runs the product calls unload() when done: the
    # product's checker would otherwise find this `resample` instead of taking its ImportError fallback)
    for name in ("soundfile", "resampy"):
        if name not in sys.modules:
            mod = types.ModuleType(name)
            mod.resample = _not_here
            mod.check_format = lambda *a, **k: True
            sys.modules[name] = mod
            _installed.append(name)
    if "statsmodels.api" not in sys.modules:
        pkg = types.ModuleType("statsmodels")
        api = types.ModuleType("statsmodels.api")
        api.nonparametric = types.SimpleNamespace(lowess=lowess)
        pkg.api = api
        sys.modules["statsmodels"] = pkg
        sys.modules["statsmodels.api"] = api
        _installed.extend(["statsmodels", "statsmodels.api"])
    if "matchering" in sys.modules and not getattr(sys.modules["matchering"], "__file__", "").startswith(STAGED):
        raise ImportError("another `matchering` is already imported in this process")
    if STAGED not in sys.path:
        sys.path.insert(0, STAGED)
    import warnings

    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        mg = importlib.import_module("matchering")
        importlib.import_module("matchering.stages")
    return mg


if __name__ == "__main__":
    m = build(force="--force" in sys.argv)
    print("no reference tree here" if m is None else f"staged {len(m['files'])} modules under {STAGED}")
