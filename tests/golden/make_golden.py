"""Freeze outputs of the UNMODIFIED reference as golden fixtures.

Run in the build container only (needs /root/reference and, for the real
statsmodels LOWESS, /opt/conda/bin/python3.9):

    python tests/golden/make_golden.py

For every case in ``cases.py`` this script
  1. builds the float32 inputs with ``matchering_amd.synth`` (system python),
  2. runs ``/root/reference/matchering/stages.py:main`` on them in a
     ``/opt/conda/bin/python3.9`` subprocess through ``oracle/reference_runner.py``
     (compiled statsmodels 0.12.2 LOWESS = the reference with its real dependency),
  3. checks ``oracle/mastering_oracle.py`` against those outputs (<= 1e-11 abs)
     -- this is what pins the oracle -- and refuses to write fixtures otherwise,
  4. writes ``<case>.npz``: the three outputs as float32, the same outputs and
     the limiter envelopes at a sparse set of frames as float64 (edges included),
     every scalar / small vector intermediate as float64, and a sha256 of the
     input bytes so that generator drift is detected by the tests.
It also freezes one LOWESS known-answer vector from the compiled statsmodels.
"""

import hashlib
import json
import os
import subprocess
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
sys.path.insert(0, HERE)

import mastering_oracle as mo  # noqa: E402
from cases import CASES, build_inputs, oracle_params, sparse_index  # noqa: E402

PY39 = "/opt/conda/bin/python3.9"
RUNNER = os.path.join(ROOT, "oracle", "reference_runner.py")


def run_reference_py39(target, reference, cfg_kwargs, extra=None):
    with tempfile.TemporaryDirectory() as tmp:
        src, dst = os.path.join(tmp, "in.npz"), os.path.join(tmp, "out.npz")
        payload = dict(target=target, reference=reference, cfg=json.dumps(cfg_kwargs))
        payload.update(extra or {})
        np.savez(src, **payload)
        subprocess.check_call([PY39, "-W", "ignore", RUNNER, src, dst])
        z = np.load(dst, allow_pickle=False)
        return {k: z[k] for k in z.files}


def main():
    # python tests/golden/make_golden.py [case ...]: only those cases (the manifest keeps the others' entries)
    only = sys.argv[1:]
    manifest = {}
    manifest_path = os.path.join(HERE, "MANIFEST.json")
    if only and os.path.exists(manifest_path):
        with open(manifest_path) as f:
            manifest = json.load(f)
    for name, case in CASES.items():
        if only and name not in only:
            continue
        target, reference = build_inputs(case)
        ref = run_reference_py39(target, reference, case["config"])
        cfg = oracle_params(case["config"])
        trace = {}
        outs = mo.master(target, reference, cfg, True, True, True, trace=trace)
        worst = 0.0
        for key, mine in zip(("result", "result_no_limiter", "result_no_limiter_normalized"), outs):
            worst = max(worst, float(np.abs(mine - ref[key]).max()))
        worst = max(worst, float(np.abs(trace["fir_mid"] - ref["fir_mid"]).max()),
                    float(np.abs(trace["fir_side"] - ref["fir_side"]).max()))
        print(f"{name}: n={target.shape[0]} oracle-vs-reference max abs diff {worst:.3e}")
        assert worst <= case.get("oracle_tolerance", 1e-11), "oracle restatement disagrees with the reference"

        idx = sparse_index(target.shape[0])
        out = {
            "input_sha256": np.array(hashlib.sha256(target.tobytes() + reference.tobytes()).hexdigest()),
            "versions": ref["versions"],
            "sparse_index": idx,
        }
        for key in ("result", "result_no_limiter", "result_no_limiter_normalized"):
            if key != "result_no_limiter_normalized":      # = no_limiter / normalize_coefficient
                out[key + "_f32"] = ref[key].astype(np.float32)
            out[key + "_sparse"] = ref[key][idx]
        out["conv_result_sparse"] = ref["conv_result"][idx]
        out["conv_mid_sparse"] = ref["conv_mid"][idx]
        for key in ("limiter_gain_attack", "limiter_slided", "limiter_gain_release"):
            if key in ref:
                out[key + "_sparse"] = ref[key][idx]
        for key in ("final_amplitude_coefficient", "target_match_rms", "target_divisions",
                    "target_piece", "reference_match_rms", "reference_divisions",
                    "reference_piece", "target_loud_count", "reference_loud_count",
                    "rms_coefficient", "correction_coefficients", "fir_mid", "fir_side",
                    "normalize_coefficient"):
            out[key] = ref[key]
        # spectra / curves come from the pinned oracle (they are internal to get_fir)
        for ch in ("mid", "side"):
            out[f"avg_target_{ch}"] = getattr(trace[ch], "avg_target")
            out[f"avg_reference_{ch}"] = getattr(trace[ch], "avg_reference")
            out[f"curve_raw_{ch}"] = getattr(trace[ch], "raw")
            out[f"curve_smooth_{ch}"] = getattr(trace[ch], "smooth")
        out["target_rmses"] = trace["target_rmses"]
        out["reference_rmses"] = trace["reference_rmses"]
        out["limiter_active"] = np.array("limiter_gain_attack" in ref)
        np.savez_compressed(os.path.join(HERE, name + ".npz"), **out)
        manifest[name] = {"frames": int(target.shape[0]), "oracle_vs_reference_max_abs": worst,
                          "versions": json.loads(str(ref["versions"]))}

    if only:
        with open(manifest_path, "w") as f:
            json.dump(manifest, f, indent=1, sort_keys=True)
        return
    # LOWESS known-answer vector from the compiled statsmodels
    rng = np.random.RandomState(7)
    n = 8193
    y = np.exp(0.8 * np.cumsum(rng.randn(n)) / np.sqrt(n)) + 0.05 * rng.randn(n)
    t, r = build_inputs(CASES["hot_lowrate"])
    ref = run_reference_py39(t[:8000], r[:8000], CASES["hot_lowrate"]["config"],
                             extra=dict(lowess_in=y, lowess_frac=0.0375, lowess_delta=0.001))
    mine = mo.lowess_it0(y, 0.0375, 0.001)
    d = float(np.abs(mine - ref["lowess_out"]).max())
    print(f"lowess: restatement-vs-compiled max abs diff {d:.3e}")
    assert d <= 1e-11
    np.savez_compressed(os.path.join(HERE, "lowess_kat.npz"), y=y, fit=ref["lowess_out"],
                        frac=0.0375, delta=0.001)
    manifest["lowess_kat"] = {"restatement_vs_compiled_max_abs": d}
    with open(os.path.join(HERE, "MANIFEST.json"), "w") as f:
        json.dump(manifest, f, indent=1, sort_keys=True)


if __name__ == "__main__":
    main()
