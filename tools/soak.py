"""Soak: mixed shapes, configurations and lanes for a fixed wall time; every result is compared bit for bit
with the first one computed for its shape.  Looks for intermittent faults, races between handles and
state that leaks from one call into the next.

    python tools/soak.py SECONDS
"""
import os
import sys
import threading
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import numpy as np

import matchering_amd as mg
from matchering_amd import stages
from matchering_amd.device import Device
from matchering_amd.synth import make_pair

SHAPES = [(7.3, 44100, 4096), (31.0, 44100, 4096), (3.1, 96000, 16384), (12.0, 48000, 2048), (5.0, 22050, 1024),
          (64.0, 44100, 4096)]


# what a Config may vary beyond the shape: LOWESS robustness passes (no operator), second-order limiter filters
# (k_limit_general), long attack / hold times (1024-block chunks)
VARIANTS = [dict(), dict(), dict(lowess_it=2),
            dict(limiter=dict(hold_filter_order=2, release_filter_order=2)),
            dict(limiter=dict(hold_filter_order=3)),
            dict(limiter=dict(attack=8.0, hold=2.0))]


def worker(lane, deadline, stats, lock):
    dev = Device(0)
    rng = np.random.RandomState(lane)
    first = {}
    while time.time() < deadline:
        k = int(rng.randint(len(SHAPES)))
        seconds, rate, fft = SHAPES[k]
        t, r = make_pair(seconds, rate, pair=k)
        v = int(rng.randint(len(VARIANTS)))
        extra = dict(VARIANTS[v])
        if "limiter" in extra:
            extra["limiter"] = mg.LimiterConfig(**extra["limiter"])
        cfg = mg.Config(internal_sample_rate=rate, fft_size=fft, max_piece_size=min(15.0, seconds / 2.5), **extra)
        need = (bool(rng.randint(2)), True, bool(rng.randint(2)))
        out = stages.main(t, r, cfg, *need, device=dev)
        key = (k, v)
        ref = first.setdefault(key, {})
        for i, o in enumerate(out):
            if o is None:
                continue
            if i in ref:
                if not np.array_equal(ref[i], o):
                    with lock:
                        stats["mismatch"] += 1
            else:
                ref[i] = o
        with lock:
            stats["calls"] += 1
    dev.close()


def main():
    seconds = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
    deadline = time.time() + seconds
    stats, lock = {"calls": 0, "mismatch": 0}, threading.Lock()
    threads = [threading.Thread(target=worker, args=(lane, deadline, stats, lock)) for lane in range(3)]
    for th in threads:
        th.start()
    for th in threads:
        th.join()
    print("soak", seconds, "s:", stats, flush=True)
    sys.exit(1 if stats["mismatch"] else 0)


if __name__ == "__main__":
    main()
