#!/usr/bin/env python
"""Where a block of the delay-line convolution spends its time: thread 0's wall clock between the phase marks of
k_conv_delay, over the blocks of one run of config #5's workload (--wide: of k_conv_wide on the headline workload).
Needs the development build

    python -m matchering_amd.build --variant convphases -DMGX_DEV_CONV_PHASES
    MGX_LIB=$PWD/tools/variants/libmgx_convphases.so python tools/conv_delay_phases.py
"""
import ctypes
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

NAMES = ["frames + pass 0 + barrier", "middle passes (2)", "row forward + barrier", "multiply + barrier",
         "row back + inverse middle passes + barrier", "inverse pass 0 + stores + peak"]


def main():
    import matchering_amd as mg
    from matchering_amd._native import library
    from matchering_amd.device import Device
    from matchering_amd.synth import make_pair

    dev = Device(0)
    wide = "--wide" in sys.argv             # the headline workload: k_conv_wide<14> (blocks of 12288 frames)
    cfg = mg.Config() if wide else mg.Config(internal_sample_rate=96000, fft_size=16384)
    native = cfg.to_native()
    target, reference = make_pair(480.0, 44100, pair=0) if wide else make_pair(240.0, 96000, pair=0)
    n, nr = target.shape[0], reference.shape[0]
    t_dev, r_dev = dev.upload(target), dev.upload(reference)
    out = dev.alloc(n * 8)
    for _ in range(3):
        dev.master(t_dev, n, r_dev, nr, native, result=None, result_no_limiter=out, want_report=False)
    dev.synchronize()
    hop = 12288 if wide else 8192
    blocks = (n + hop - 1) // hop
    raw = np.zeros((blocks, 8), np.uint32)
    library().mgx_dev_conv_ticks_read(raw.ctypes.data_as(ctypes.c_void_p), blocks)
    t = raw[:, :6].astype(np.float64) / 100.0           # us (100 MHz wall clock)
    start = (raw[:, 7].astype(np.int64) - int(raw[:, 7].min())) % (1 << 32) / 100.0
    print(f"{blocks} blocks; thread 0's wall clock between marks, us: mean / median / 90th percentile")
    for k, name in enumerate(NAMES):
        print(f"  {name:44s} {t[:, k].mean():8.2f} {np.median(t[:, k]):8.2f} {np.percentile(t[:, k], 90):8.2f}")
    total = t.sum(axis=1)
    print(f"  {'whole block':44s} {total.mean():8.2f} {np.median(total):8.2f} {np.percentile(total, 90):8.2f}")
    print(f"  first start to last end: {(start + total).max():.1f} us")


if __name__ == "__main__":
    main()
