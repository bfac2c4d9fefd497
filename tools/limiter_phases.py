#!/usr/bin/env python
"""Where a limiter chunk's time goes: thread 0's wall-clock time between the phase marks of limit_chunk,
averaged over the chunks of one run.  Needs the development build

    python -m matchering_amd.build --variant phases -DMGX_DEV_LIMITER_PHASES
    MGX_LIB=$PWD/tools/variants/libmgx_phases.so python tools/limiter_phases.py
"""
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

NAMES = ["load", "hold window + scan", "attack window + scan", "attack forward / scan / backward",
         "take hold (wave 0)", "barrier (attack take, wave 1)", "hold output + release scan",
         "publish, ask, reload issue", "take release", "gain", "store"]


def main():
    import matchering_amd as mg
    from matchering_amd._native import library
    from matchering_amd.device import Device
    from matchering_amd.synth import make_pair

    dev = Device(0)
    cfg = mg.Config()
    native = cfg.to_native()
    target, reference = make_pair(480.0, 44100, pair=0)
    n, nr = target.shape[0], reference.shape[0]
    t_dev, r_dev = dev.upload(target), dev.upload(reference)
    out = dev.alloc(n * 8)
    lib = library()
    import numpy as np
    for _ in range(3):
        dev.master(t_dev, n, r_dev, nr, native, result=out, want_report=False)
    dev.synchronize()
    cap = 16384
    raw = np.zeros((cap, 16), np.uint32)
    lib.mgx_dev_phase_ticks_read(raw.ctypes.data_as(ctypes.c_void_p), cap)
    chunks = int(np.count_nonzero(raw[:, 10]))
    t = raw[:chunks].astype(np.float64) / 100.0           # us
    start = (raw[:chunks, 15].astype(np.int64) - int(raw[:chunks, 15].min())) % (1 << 32) / 100.0
    print(f"{chunks} chunks; thread 0's wall clock between marks, us: mean / median / 90th percentile")
    for k, name in enumerate(NAMES):
        print(f"  {name:36s} {t[:, k].mean():8.2f} {np.median(t[:, k]):8.2f} {np.percentile(t[:, k], 90):8.2f}")
    total = t[:, :11].sum(axis=1)
    print(f"  {'whole chunk':36s} {total.mean():8.2f} {np.median(total):8.2f} {np.percentile(total, 90):8.2f}")
    end = start + total
    print(f"  first start to last end {end.max():.1f} us; chunk-time / that = {total.sum() / end.max():.0f} chunks in flight on average")
    order = np.argsort(start)
    print("  start time of chunk c minus start time of chunk c-1, us: mean %.3f, min %.2f, max %.2f; chunks started before their predecessor: %d"
          % (np.diff(start).mean(), np.diff(start).min(), np.diff(start).max(), int((np.diff(start) < 0).sum())))
    # who is late: chunk c takes its release carry at take[c]; predecessor c-k published its word at pub[c-k]
    pub = start + t[:, :7].sum(axis=1)
    take = start + t[:, :8].sum(axis=1)
    reach = 155
    late_by = np.full(chunks, -1e9)
    who = np.zeros(chunks, np.int64)
    for k in range(1, reach + 1):
        d = np.full(chunks, -1e9)
        d[k:] = pub[:-k] - take[k:]
        better = d > late_by
        late_by[better] = d[better]
        who[better] = k
    waited = t[:, 8]
    print("  release carry: latest predecessor's publish time minus own arrival at the take, us: mean %.2f, median %.2f, 90th %.2f"
          % (late_by[reach:].mean(), np.median(late_by[reach:]), np.percentile(late_by[reach:], 90)))
    print("  measured wait minus max(0, that): mean %.2f us (hand-off + poll granularity)" % (waited[reach:] - np.maximum(0, late_by[reach:])).mean())
    hist = np.bincount(np.minimum(who[reach:], 64), minlength=65)
    print("  distance of the latest predecessor: 1: %d, 2-4: %d, 5-16: %d, 17-63: %d, 64+: %d chunks"
          % (hist[1], hist[2:5].sum(), hist[5:17].sum(), hist[17:64].sum(), hist[64]))
    dur_to_pub = t[:, :7].sum(axis=1)
    print("  start -> release publish, us: mean %.2f, sd %.2f, 99th %.2f" % (dur_to_pub.mean(), dur_to_pub.std(), np.percentile(dur_to_pub, 99)))
    for lo in range(0, chunks, max(1, chunks // 12)):
        hi = min(chunks, lo + max(1, chunks // 12))
        print(f"  chunks {lo:5d}..{hi:5d}: start {start[lo:hi].min():7.1f}..{start[lo:hi].max():7.1f}  take-hold {t[lo:hi, 4].mean():6.2f}  take-release {t[lo:hi, 8].mean():6.2f}  whole {total[lo:hi].mean():6.2f}")


if __name__ == "__main__":
    main()
