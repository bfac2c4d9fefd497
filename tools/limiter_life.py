#!/usr/bin/env python
"""How many limiter chunks are alive at a time, and for how long: every chunk of one k_limit launch stamps its kernel
entry, the end of its load and its end (development build, -DMGX_DEV_LIMITER_PHASES).

    python -m matchering_amd.build --variant phases -DMGX_DEV_LIMITER_PHASES
    MGX_LIB=$PWD/tools/variants/libmgx_phases.so python tools/limiter_life.py
"""
import ctypes
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import matchering_amd as mg
    from matchering_amd._native import library
    from matchering_amd.device import Device
    from matchering_amd.synth import make_pair

    dev = Device(0)
    native = mg.Config().to_native()
    target, reference = make_pair(480.0, 44100, pair=0)
    n, nr = target.shape[0], reference.shape[0]
    t_dev, r_dev = dev.upload(target), dev.upload(reference)
    out = dev.alloc(n * 8)
    for _ in range(3):
        dev.master(t_dev, n, r_dev, nr, native, result=out, want_report=False)
    dev.synchronize()
    lib = ctypes.CDLL(os.environ["MGX_LIB"])
    cap = 16384
    raw = np.zeros((cap, 8), np.int64)
    lib.mgx_dev_chunk_life_read(raw.ctypes.data_as(ctypes.c_void_p), cap)
    chunks = int(np.count_nonzero(raw[:, 2]))
    life = raw[:chunks]
    t0 = life[:, 0].min()
    entry, loaded, end = ((life[:, k] - t0) / 100.0 for k in range(3))          # us (100 MHz wall clock)
    kind = life[:, 3] & 0xff
    hw = (life[:, 3] >> 8) & 0xffffff
    xcc = life[:, 3] >> 32
    total = end.max()
    print(f"{chunks} chunks ({(kind == 1).sum()} busy, {(kind == 2).sum()} quiet, {(kind == 0).sum()} at the ends); first entry to last end {total:.1f} us")
    for name, k in (("busy", 1), ("quiet", 2), ("end", 0)):
        sel = kind == k
        if not sel.any():
            continue
        d = end[sel] - entry[sel]
        ld = loaded[sel] - entry[sel]
        print(f"  {name:6s} lifetime mean {d.mean():6.2f} median {np.median(d):6.2f} 90th {np.percentile(d, 90):6.2f} us;"
              f" of which entry -> loaded mean {ld[ld > 0].mean() if (ld > 0).any() else 0:5.2f}")
    # alive at time t
    grid = np.arange(0.0, total, 2.0)
    alive = [(int(((entry <= t) & (end > t)).sum()), int(((entry <= t) & (end > t) & (kind == 1)).sum())) for t in grid]
    print("  alive (all / busy) every 10 us: " + " ".join(f"{a}/{b}" for a, b in alive[::5]))
    print(f"  mean alive {np.mean([a for a, _ in alive]):.0f} of 1024 slots; sum of lifetimes / span = {(end - entry).sum() / total:.0f}")
    # entries per microsecond: is the dispatch rate the limit?
    order = np.sort(entry)
    print(f"  entries: first 1024 within {order[min(1023, chunks - 1)]:.1f} us; then {chunks - 1024} more in {order[-1] - order[min(1023, chunks - 1)]:.1f} us"
          f" = {max(0, chunks - 1024) / max(1e-9, order[-1] - order[min(1023, chunks - 1)]):.1f} per us")
    # chunk number against entry time (tickets are drawn at entry: order must hold) and against end time
    print(f"  chunks whose entry precedes their predecessor's: {(np.diff(entry) < 0).sum()}; end order inversions beyond 100 chunks: "
          f"{sum(1 for c in range(100, chunks) if end[c] < end[c - 100])}")
    # the slot view: how soon after a chunk ends does the next entry happen on the same compute unit?
    key = ((hw.astype(np.int64) >> 8) & 0xff) * 16 + xcc           # cu_id, sh_id, se_id of HW_ID + the XCC
    gaps = []
    for k in np.unique(key):
        idx = np.where(key == k)[0]
        e = np.sort(end[idx])
        s = np.sort(entry[idx])
        # greedy: each entry after the first four matches the earliest unmatched end
        for i in range(4, len(s)):
            gaps.append(s[i] - e[i - 4])
    if gaps:
        gaps = np.array(gaps)
        print(f"  compute-unit slots ({len(np.unique(key))} distinct units seen): entry of the (i+4)th chunk minus end of the ith on the same unit: "
              f"mean {gaps.mean():.2f} median {np.median(gaps):.2f} 90th {np.percentile(gaps, 90):.2f} us")
    # the critical path: when are a chunk's words out, when has it got its predecessors'?
    stamps = {name: (life[:, k] - t0) / 100.0 - entry for name, k in (("hold word out", 4), ("hold carry in", 5),
                                                                        ("release word out", 6), ("release carry in", 7))}
    for name, k in (("busy", 1), ("quiet", 2)):
        sel = kind == k
        print(f"  {name:6s} after entry, mean us: loaded {(loaded - entry)[sel].mean():5.2f}  " +
              "  ".join(f"{n} {v[sel].mean():5.2f}" for n, v in stamps.items()) + f"  end {(end - entry)[sel].mean():5.2f}")
    # the release take: when did the latest of the 155 predecessors publish, relative to this chunk's own publish?
    rel_out = (life[:, 6] - t0) / 100.0
    rel_in = (life[:, 7] - t0) / 100.0
    hold_out = (life[:, 4] - t0) / 100.0
    hold_in = (life[:, 5] - t0) / 100.0
    late_r, late_h = [], []
    for c in range(160, chunks):
        late_r.append(rel_out[c - 155:c].max() - rel_out[c])
        late_h.append(hold_out[c - 6:c].max() - hold_out[c])
    late_r, late_h = np.array(late_r), np.array(late_h)
    wait_r = (rel_in - rel_out)[160:]
    wait_h = (hold_in - hold_out)[160:]
    print(f"  release: latest predecessor's word minus own word: mean {late_r.mean():5.2f} median {np.median(late_r):5.2f} 90th {np.percentile(late_r, 90):5.2f};"
          f"  own word out -> carry in: mean {wait_r.mean():5.2f}; of that beyond the latest predecessor: {(wait_r - np.maximum(late_r, 0)).mean():5.2f}")
    print(f"  hold:    latest predecessor's word minus own word: mean {late_h.mean():5.2f} median {np.median(late_h):5.2f} 90th {np.percentile(late_h, 90):5.2f};"
          f"  own word out -> carry in: mean {wait_h.mean():5.2f}; of that beyond the latest predecessor: {(wait_h - np.maximum(late_h, 0)).mean():5.2f}")
    per_unit = np.bincount(np.unique(key, return_inverse=True)[1])
    print(f"  chunks per unit: min {per_unit.min()} mean {per_unit.mean():.1f} max {per_unit.max()}")


if __name__ == "__main__":
    main()
