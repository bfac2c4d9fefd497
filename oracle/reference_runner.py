"""Run the UNMODIFIED reference (``/root/reference/matchering``) as the pinning
authority for the oracle -- TEST INFRASTRUCTURE, build container only.

``/root/reference`` does not exist on the GPU box; nothing under ``tests/ -m gpu``,
``smoke()`` or ``bench.py`` imports this file.  It is used by
``tests/golden/make_golden.py`` (to freeze reference outputs as fixtures) and by
the optional ``tests/test_oracle_vs_reference.py`` (skipped when the reference
tree is absent).

The reference imports three packages that are not installed for the system
interpreter.  Two are file I/O only and never reached from ``stages.main``
(``soundfile``: loader.py:23, saver.py:22, results.py:22; ``resampy``:
checker.py:22) -- they get empty stub modules.  The third is numerical:
``statsmodels.api`` (dsp.py:22), used at one call site, ``dsp.py:103-106``
``sm.nonparametric.lowess``.  Two ways of supplying it:

* under ``/opt/conda/bin/python3.9`` the compiled statsmodels 0.12.2 LOWESS is
  importable (``statsmodels.nonparametric.smoothers_lowess``; the package's
  ``statsmodels.api`` itself is broken there by a numpy incompatibility) and is
  wired in as ``sm.nonparametric.lowess`` -- this is "the reference with its
  real dependency";
* under the system interpreter, where statsmodels is absent, the oracle's
  ``lowess_it0`` restatement is wired in (only after the golden script has
  shown it equal to the compiled one).

Command line (used from make_golden.py under python3.9):
    reference_runner.py IN.npz OUT.npz
IN.npz holds target, reference (float32 (n,2)) and ``cfg`` = JSON of Config
kwargs (+ "limiter": LimiterConfig kwargs); OUT.npz receives the three outputs
of ``stages.main`` and the intermediates captured by wrapping stage helpers.
"""

import importlib
import json
import os
import sys
import types

REFERENCE_ROOT = "/root/reference"


def reference_available():
    return os.path.isdir(os.path.join(REFERENCE_ROOT, "matchering"))


def _install_stubs(lowess_impl):
    for name in ("soundfile", "resampy"):
        if name not in sys.modules:
            mod = types.ModuleType(name)
            mod.resample = None
            mod.check_format = lambda *a, **k: True
            sys.modules[name] = mod
    sm_pkg = types.ModuleType("statsmodels")
    sm_api = types.ModuleType("statsmodels.api")
    sm_api.nonparametric = types.SimpleNamespace(lowess=lowess_impl)
    sm_pkg.api = sm_api
    sys.modules["statsmodels.api"] = sm_api
    # keep a real 'statsmodels' package if one was already imported (python3.9 path)
    sys.modules.setdefault("statsmodels", sm_pkg)
    setattr(sys.modules["statsmodels"], "api", sm_api)


def compiled_lowess():
    """The compiled statsmodels LOWESS, or None when statsmodels is absent."""
    try:
        from statsmodels.nonparametric.smoothers_lowess import lowess
        return lowess
    except Exception:
        return None


def load_reference(lowess_impl=None):
    """Import and return the reference package with stubs in place."""
    import numpy as np

    if lowess_impl is None:
        lowess_impl = compiled_lowess()
    if lowess_impl is None:
        sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
        from mastering_oracle import lowess_it0

        def lowess_impl(endog, exog, frac, it, delta):
            assert it == 0
            assert np.array_equal(exog, np.linspace(0, 1, len(endog)))
            return np.stack((exog, lowess_it0(endog, frac, delta)), axis=1)

    _install_stubs(lowess_impl)
    if REFERENCE_ROOT not in sys.path:
        sys.path.insert(0, REFERENCE_ROOT)
    import warnings

    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        return importlib.import_module("matchering")


def make_config(mg, cfg_kwargs):
    kw = dict(cfg_kwargs)
    lim = kw.pop("limiter", None)
    if lim is not None:
        kw["limiter"] = mg.defaults.LimiterConfig(**lim)
    return mg.Config(**kw)


def run_reference(target, reference, cfg_kwargs, need=(True, True, True), capture=True):
    """stages.main of the reference on float64 copies of the inputs.  Returns
    (outputs tuple, dict of intermediates)."""
    import numpy as np

    mg = load_reference()
    stages = importlib.import_module("matchering.stages")
    cfg = make_config(mg, cfg_kwargs)
    trace = {}
    originals = {}
    if capture:
        # wrap the helpers *as seen from stages.py* to record what flows between stages
        def wrap(name, fn):
            def inner(*a, **k):
                out = fn(*a, **k)
                trace.setdefault(name, []).append(out)
                return out
            return inner

        for name in ("normalize_reference", "analyze_levels", "get_fir", "convolve",
                     "get_average_rms", "get_lpis_and_match_rms",
                     "get_rms_c_and_amplify_pair", "limit", "normalize"):
            originals[name] = getattr(stages, name)
            setattr(stages, name, wrap(name, originals[name]))
        # the limiter's internals live in hyrax.py as module-level helpers (hyrax.py:43-75)
        hyrax = importlib.import_module("matchering.limiter.hyrax")
        hyrax_originals = {}
        for name in ("__process_attack", "__process_release"):
            hyrax_originals[name] = getattr(hyrax, name)
            setattr(hyrax, name, wrap(name, hyrax_originals[name]))
    try:
        import warnings

        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            outs = stages.main(
                np.array(target, dtype=np.float64), np.array(reference, dtype=np.float64),
                cfg, need_default=need[0], need_no_limiter=need[1],
                need_no_limiter_normalized=need[2])
    finally:
        for name, fn in originals.items():
            setattr(stages, name, fn)
        if capture:
            for name, fn in hyrax_originals.items():
                setattr(hyrax, name, fn)
    inter = {}
    if capture:
        inter["final_amplitude_coefficient"] = trace["normalize_reference"][0][1]
        t, r = trace["analyze_levels"][0], trace["analyze_levels"][1]
        inter["target_match_rms"], inter["target_divisions"], inter["target_piece"] = t[4], t[5], t[6]
        inter["reference_match_rms"], inter["reference_divisions"], inter["reference_piece"] = r[4], r[5], r[6]
        inter["target_loud_count"] = t[2].shape[0]
        inter["reference_loud_count"] = r[2].shape[0]
        pairs = trace["get_rms_c_and_amplify_pair"]
        inter["rms_coefficient"] = pairs[0][0]
        inter["correction_coefficients"] = np.array([p[0] for p in pairs[1:]])
        inter["fir_mid"], inter["fir_side"] = trace["get_fir"][0], trace["get_fir"][1]
        inter["conv_result"], inter["conv_mid"] = (np.ascontiguousarray(trace["convolve"][0][0]),
                                                   trace["convolve"][0][1])
        if "normalize" in trace:
            inter["normalize_coefficient"] = trace["normalize"][0][1]
        if "__process_attack" in trace:
            inter["limiter_gain_attack"], inter["limiter_slided"] = trace["__process_attack"][0]
            inter["limiter_gain_release"] = trace["__process_release"][0]
    return outs, inter


def _main(argv):
    import numpy as np

    src, dst = argv[1], argv[2]
    z = np.load(src, allow_pickle=False)
    cfg_kwargs = json.loads(str(z["cfg"]))
    outs, inter = run_reference(z["target"], z["reference"], cfg_kwargs)
    payload = {"result": np.ascontiguousarray(outs[0]),
               "result_no_limiter": np.ascontiguousarray(outs[1]),
               "result_no_limiter_normalized": np.ascontiguousarray(outs[2])}
    payload.update({k: np.asarray(v) for k, v in inter.items()})
    if "lowess_in" in z.files:
        lw = compiled_lowess()
        y = z["lowess_in"]
        payload["lowess_out"] = lw(y, np.linspace(0, 1, len(y)), frac=float(z["lowess_frac"]),
                                   it=0, delta=float(z["lowess_delta"]))[:, 1]
        payload["lowess_impl"] = np.array("statsmodels-compiled")
    import numpy, scipy
    payload["versions"] = np.array(json.dumps({
        "python": sys.version.split()[0], "numpy": numpy.__version__, "scipy": scipy.__version__,
        "lowess": "compiled statsmodels" if compiled_lowess() else "oracle restatement"}))
    np.savez(dst, **payload)


if __name__ == "__main__":
    _main(sys.argv)
