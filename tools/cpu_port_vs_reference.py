#!/usr/bin/env python
"""Wall time of the oracle (oracle/mastering_oracle.py, the CPU baseline bench.py can run on a GPU box)
against the UNMODIFIED reference (/root/reference/matchering, build container only) on the same pair.

    python tools/cpu_port_vs_reference.py [--seconds 120]  ->  profiles/cpu_port_vs_reference.json

bench.py copies the file into its ``cpu_baseline`` object so that the reader can translate the port's
number into the reference's.  The reference runs with its I/O imports stubbed and the oracle's LOWESS
restatement in place of statsmodels (oracle/reference_runner.py), numpy/scipy as installed here.
"""
import argparse
import json
import os
import platform
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))


def best_of(fn, runs):
    out = []
    for _ in range(runs):
        t0 = time.perf_counter()
        fn()
        out.append(time.perf_counter() - t0)
    return min(out)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seconds", type=float, default=120.0)
    ap.add_argument("--runs", type=int, default=3)
    args = ap.parse_args()
    import numpy
    import scipy

    import mastering_oracle as mo
    import reference_runner as rr
    from matchering_amd.synth import make_pair

    if not rr.reference_available():
        raise SystemExit("/root/reference is not present: run this on the build container")
    target, reference = make_pair(args.seconds, 44100, pair=0)
    ocfg = mo.params()
    result = {"pair": f"{args.seconds:.0f} s stereo 44.1 kHz synthetic pair ({target.shape[0]} frames), best of 2 blocks of {args.runs} consecutive runs each",
              "host": f"{platform.processor() or platform.machine()}, {os.cpu_count()} logical cores (build container)",
              "versions": {"python": platform.python_version(), "numpy": numpy.__version__, "scipy": scipy.__version__}}
    for name, need in (("full", (True, False, False)), ("fir_only", (False, True, False))):
        # Blocks of consecutive runs of one implementation (a run that follows a different code path pays
        # for a cold allocator: +30..40 % here, for either of them), the two blocks repeated twice
        t_port = t_ref = float("inf")
        for _ in range(2):
            t_port = min(t_port, best_of(lambda: mo.master(target, reference, ocfg, *need), args.runs))
            t_ref = min(t_ref, best_of(lambda: rr.run_reference(target, reference, {}, need=need, capture=False), args.runs))
        # ... and what the two compute: the largest difference between their outputs on this pair (the oracle's pin at
        # this size; the frozen fixtures of tests/golden are <= 307 k frames)
        import numpy as np

        got = [o for o in mo.master(target, reference, ocfg, *need) if o is not None][0]
        ref_outs, _ = rr.run_reference(target, reference, {}, need=need, capture=False)
        want = [o for o in ref_outs if o is not None][0]
        result[name] = {"oracle_s": round(t_port, 3), "reference_s": round(t_ref, 3),
                        "oracle_over_reference": round(t_port / t_ref, 3),
                        "max_abs_difference_of_outputs": float(np.abs(np.asarray(got) - np.asarray(want)).max())}
        print(name, result[name], flush=True)
    path = os.path.join(ROOT, "profiles", "cpu_port_vs_reference.json")
    try:                                     # (the GPU host's own pair of numbers, recorded once, stays in the file)
        with open(path) as fh:
            kept = json.load(fh).get("on_the_gpu_box_host")
        if kept:
            result["on_the_gpu_box_host"] = kept
    except (OSError, ValueError):
        pass
    with open(path, "w") as fh:
        json.dump(result, fh, indent=1)


if __name__ == "__main__":
    main()
