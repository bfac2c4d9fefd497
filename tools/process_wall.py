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
    ap.add_argument("--batch", type=int, default=0,
                    help="instead: this many pairs through batch.process_batch (three lanes, four I/O threads)")
    args = ap.parse_args()
    if args.batch:
        return batch_wall(args)

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


def batch_wall(args):
    """Files to files for a batch: loaders, three lanes on the GPU and writers overlap."""
    import matchering_amd as mg
    from matchering_amd import audio_io, batch
    from matchering_amd.synth import make_pair

    folder = tempfile.mkdtemp(prefix="mgx_wall_")
    jobs, frames = [], 0
    for k in range(args.batch):
        target, reference = make_pair(args.seconds, 44100, pair=k % 4)
        tp, rp, op = (os.path.join(folder, f"{n}{k}.wav") for n in ("target", "reference", "result"))
        audio_io.write_wav(tp, target * 0.7, 44100, args.subtype)
        audio_io.write_wav(rp, reference * 0.9, 44100, args.subtype)
        jobs.append({"target": tp, "reference": rp, "results": [mg.Result(op, args.subtype)]})
        frames += target.shape[0]
    print(f"{args.batch} pairs of {args.seconds:.0f} s at 44.1 kHz, {args.subtype} in, {args.subtype} out")
    for run in range(args.runs):
        t0 = time.perf_counter()
        batch.process_batch(jobs, mg.Config(), rank=0, world_size=1)
        total = time.perf_counter() - t0
        print(f"run {run}: total {total * 1e3:8.1f} ms = {frames / total / 1e6:7.1f} M frames/s "
              f"({total / args.batch * 1e3:.1f} ms per pair)")
    import shutil

    shutil.rmtree(folder, ignore_errors=True)


if __name__ == "__main__":
    main()
