#!/usr/bin/env python
"""Where do k_correction_round's last workgroup and k_correction_tail spend their time?

Needs the trace variant of the library (python -m matchering_amd.build --variant tailtrace -DMGX_TAIL_TRACE)
selected with MGX_LIB; prints, for the last of a few mgx_master calls on the 8-minute pair, the 100 MHz
phase stamps of the two kernels relative to the first workgroup's entry into k_correction_tail.
"""
import ctypes
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import matchering_amd as mg
    from matchering_amd import _native
    from matchering_amd.device import Device
    from matchering_amd.synth import make_pair

    dev = Device(0)
    cfg = mg.Config()
    native = cfg.to_native()
    target, reference = make_pair(480.0, 44100, pair=0)
    n, nr = target.shape[0], reference.shape[0]
    t_dev, r_dev = dev.upload(target), dev.upload(reference)
    out = dev.alloc(n * 8)
    lib = _native.library()
    fn = lib.mgx_debug_tail_trace
    fn.argtypes = [ctypes.c_void_p, ctypes.c_void_p]
    tail = np.zeros((160, 32), np.uint64)
    rnd = np.zeros(8, np.uint64)
    for it in range(6):
        dev.master(t_dev, n, r_dev, nr, native, result=out, want_report=False)
        dev.synchronize()
        assert fn(tail.ctypes.data, rnd.ctypes.data) == 0
        used = tail[:, 0] > 0
        t = tail[used].astype(np.int64)
        t0 = t[:, 0].min()
        us = lambda v: (np.asarray(v, np.int64) - t0) / 100.0
        print(f"-- call {it}: {used.sum()} workgroups in the tail; k_correction_round: entry {us(rnd[0]):.1f}, last arriver "
              f"found at {us(rnd[1]):.1f}, decided at {us(rnd[2]):.1f} us (0 = first tail workgroup's entry)")
        names = ["entry", "published preset", "lists cached"]
        for r in range(3):
            names += [f"r{r + 1} summed", f"r{r + 1} partial out", f"r{r + 1} arrived", f"r{r + 1} decider read", f"r{r + 1} decider done",
                      f"r{r + 1} gain seen"]
        for slot, name in enumerate(names):
            col = t[:, slot]
            ok = col >= t0
            if not ok.any():
                continue
            v = us(col[ok])
            print(f"   {name:18s} n={ok.sum():3d}  min {v.min():7.2f}  median {np.median(v):7.2f}  max {v.max():7.2f}")


if __name__ == "__main__":
    main()
