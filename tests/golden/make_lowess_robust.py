#!/usr/bin/env python
"""Known-answer vectors of LOWESS WITH robustness iterations from the compiled statsmodels the reference
depends on (smoothers_lowess.lowess, statsmodels 0.12.2, run through /opt/conda/bin/python3.9 -- the only
interpreter of this image that has it): ``tests/golden/lowess_robust_kat.npz``.

    python tests/golden/make_lowess_robust.py

The reference reaches it through ``dsp.smooth_lowess`` (dsp.py:103-106) with ``Config.lowess_it``
(defaults.py:76, default 0).  The input carries two outliers so that the robustness weights matter.
"""
import os
import subprocess
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
PY39 = "/opt/conda/bin/python3.9"

CHILD = r"""
import sys, numpy as np
from statsmodels.nonparametric.smoothers_lowess import lowess
import statsmodels
z = np.load(sys.argv[1])
y = z["y"]; x = np.linspace(0, 1, y.shape[0])
out = {"y": y, "frac": z["frac"], "delta": z["delta"], "statsmodels": np.array(statsmodels.__version__)}
for it in (1, 2, 3):
    out["fit%d" % it] = lowess(y, x, frac=float(z["frac"]), it=it, delta=float(z["delta"]))[:, 1]
np.savez_compressed(sys.argv[2], **out)
"""


def main():
    rng = np.random.RandomState(7)
    n = 8193
    y = np.exp(0.8 * np.cumsum(rng.randn(n)) / np.sqrt(n)) + 0.05 * rng.randn(n)
    y[1000] += 3.0
    y[4000:4010] -= 2.0
    with tempfile.TemporaryDirectory() as tmp:
        src, dst = os.path.join(tmp, "in.npz"), os.path.join(HERE, "lowess_robust_kat.npz")
        np.savez(src, y=y, frac=0.0375, delta=0.001)
        subprocess.check_call([PY39, "-W", "ignore", "-c", CHILD, src, dst])
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(HERE)), "oracle"))
    import mastering_oracle as mo

    z = np.load(dst)
    for it in (1, 2, 3):
        d = float(np.abs(mo.lowess(z["y"], 0.0375, 0.001, it) - z[f"fit{it}"]).max())
        print(f"it={it}: restatement-vs-compiled max abs diff {d:.3e}")
        assert d <= 1e-11


if __name__ == "__main__":
    main()
