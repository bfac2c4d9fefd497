#!/usr/bin/env python
"""Wall time of ``mg.process`` file to file (core.py:32-121) on one pair, with the share of each step:

    python tools/process_wall.py [--seconds 480] [--subtype PCM_16] [--runs 3]

Writes a synthetic target / reference pair as WAV of the given subtype to a temporary folder, then runs
``mg.process(target, reference, [Result(out, subtype)])`` and reports, per run, the time inside load, check,
stages.main and save (host steps around the one call that runs on the GPU) and the total.
"""
import argparse
import os
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seconds", type=float, default=480.0)
    ap.add_argument("--subtype", default="PCM_16")
    ap.add_argument("--runs", type=int, default=3)
    args = ap.parse_args()

    import matchering_amd as mg
    from matchering_amd import audio_io, core
    from matchering_amd.synth import make_pair

    target, reference = make_pair(args.seconds, 44100, pair=0)
    folder = tempfile.mkdtemp(prefix="mgx_wall_")
    tp, rp, op = (os.path.join(folder, n) for n in ("target.wav", "reference.wav", "result.wav"))
    audio_io.write_wav(tp, target * 0.7, 44100, args.subtype)
    audio_io.write_wav(rp, reference * 0.9, 44100, args.subtype)
    frames = target.shape[0]
    del target, reference

    spent = {}

    def timed(module, name, label):
        inner = getattr(module, name)

        def wrapper(*a, **k):
            t0 = time.perf_counter()
            try:
                return inner(*a, **k)
            finally:
                spent[label] = spent.get(label, 0.0) + time.perf_counter() - t0

        setattr(module, name, wrapper)

    for name, label in (("load", "load"), ("check", "check"), ("check_equality", "check_equality"), ("main", "stages.main"),
                        ("save", "save")):
        timed(core, name, label)

    print(f"{frames} frames per track ({args.seconds:.0f} s at 44.1 kHz), {args.subtype} in, {args.subtype} out")
    for run in range(args.runs):
        spent.clear()
        t0 = time.perf_counter()
        mg.process(tp, rp, [mg.Result(op, args.subtype)])
        total = time.perf_counter() - t0
        parts = "  ".join(f"{k} {v * 1e3:8.1f} ms" for k, v in spent.items())
        print(f"run {run}: total {total * 1e3:8.1f} ms = {frames / total / 1e6:7.1f} M frames/s   [{parts}]")
    for p in (tp, rp, op):
        os.remove(p)
    os.rmdir(folder)


if __name__ == "__main__":
    main()
