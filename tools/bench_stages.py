#!/usr/bin/env python
"""Interleaved A/B of library environment switches inside ONE process (one box, one clock state).

    python tools/bench_stages.py [--seconds 480] [--rounds 7] [--fir-only] VARIANT [VARIANT ...]

A VARIANT is ``name`` or ``name:VAR=value,VAR=value``; ``base`` sets nothing.  Every round runs each
variant once (mgx_master on the resident synthetic pair, stage timing on); the table shows the
median device time per stage (HIP events on the handle's stream) and the host wall time per call.
The first variant's result is the yardstick the others are compared with (max |difference|).
"""
import argparse
import os
import statistics
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("variants", nargs="+")
    ap.add_argument("--seconds", type=float, default=480.0)
    ap.add_argument("--sample-rate", type=int, default=44100)
    ap.add_argument("--fft-size", type=int, default=4096)
    ap.add_argument("--rounds", type=int, default=7)
    ap.add_argument("--fir-only", action="store_true")
    ap.add_argument("--preheat", type=float, default=0.0,
                    help="seconds of all-SIMD FMA load (tools/mgx_probe.py) right before the timed rounds: does the box "
                         "need waking up?")
    args = ap.parse_args()

    import matchering_amd as mg
    from matchering_amd._native import STAGES
    from matchering_amd.device import Device
    from matchering_amd.synth import make_pair

    variants = []
    for v in args.variants:
        name, _, env = v.partition(":")
        variants.append((name, dict(kv.split("=", 1) for kv in env.split(",") if kv)))
    touched = sorted({k for _, env in variants for k in env})

    dev = Device(0)
    cfg = mg.Config(internal_sample_rate=args.sample_rate, fft_size=args.fft_size)
    native = cfg.to_native()
    target, reference = make_pair(args.seconds, args.sample_rate, pair=0)
    n, nr = target.shape[0], reference.shape[0]
    t_dev, r_dev = dev.upload(target), dev.upload(reference)
    out = dev.alloc(n * 8)
    dev.stage_timing(True)

    def run():
        if args.fir_only:
            dev.master(t_dev, n, r_dev, nr, native, result=None, result_no_limiter=out, want_report=False)
        else:
            dev.master(t_dev, n, r_dev, nr, native, result=out, want_report=False)

    def apply(env):
        for k in touched:
            os.environ.pop(k, None)
        os.environ.update(env)

    times = {name: {s: [] for s in STAGES} for name, _ in variants}
    walls = {name: [] for name, _ in variants}
    yardstick = None
    for name, env in variants:          # warm-up + correctness against the first variant
        apply(env)
        run()
        dev.synchronize()
        got = dev.download(out, (n, 2))
        if yardstick is None:
            yardstick = got
            print(f"{name:12s} yardstick, peak {np.abs(got).max():.6f}")
        else:
            print(f"{name:12s} max |diff| vs {variants[0][0]}: {np.abs(got - yardstick).max():.3e}")
    if args.preheat > 0:
        import mgx_probe        # tools/mgx_probe.py (this folder is on sys.path: the script lives in it)

        res = [0.0] * 4
        t_end = time.perf_counter() + args.preheat
        while time.perf_counter() < t_end:
            res = mgx_probe.clock(dev.index, 8192, 600000)
        print(f"preheated {args.preheat:.1f} s: shader {res[2]:.0f} MHz under load")
    for _ in range(args.rounds):
        for name, env in variants:
            apply(env)
            dev.synchronize()
            t0 = time.perf_counter()
            run()
            dev.synchronize()
            walls[name].append((time.perf_counter() - t0) * 1e3)
            for s, v in dev.stage_times().items():
                if v is not None:
                    times[name][s].append(v * 1e3)
    shown = [s for s in STAGES if any(times[name][s] for name, _ in variants)]
    print(f"{'variant':12s}" + "".join(f"{s[:14]:>15s}" for s in shown) + f"{'sum':>10s}{'wall':>10s}   (us, median of {args.rounds})")
    for name, _ in variants:
        med = [statistics.median(times[name][s]) if times[name][s] else float('nan') for s in shown]
        print(f"{name:12s}" + "".join(f"{m:15.1f}" for m in med) + f"{np.nansum(med):10.1f}{statistics.median(walls[name]) * 1e3:10.1f}")


if __name__ == "__main__":
    main()
