"""Phase timeline of k_conv from a stamps dump (library built with -DMGX_CONV_STAMPS, run with
MGX_STAMPS_OUT=<file>): median / p10 / p90 cycles of every phase over workgroups and pairs.

    python tools/conv_stamps.py stamps.bin
"""
import sys

import numpy as np

NAMES = ["load+pass0 (mid)", "barrier", "fwd middle", "barrier", "row: fwd-last x H inv-first", "barrier",
         "inv middle", "barrier", "inv pass0 -> keep mid", "barrier",
         "load+pass0 (side)", "barrier", "fwd middle", "barrier", "row", "barrier", "inv middle", "barrier",
         "inv pass0 + store", "peak reduce + barrier"]

raw = np.fromfile(sys.argv[1], dtype=np.int64).reshape(-1, 8, 32)
rows = raw.reshape(-1, 32)
rows = rows[(rows[:, 0] != 0) & (rows[:, 20] != 0)]
d = np.diff(rows[:, :21], axis=1).astype(np.float64)
total = rows[:, 20] - rows[:, 0]
print(f"{len(rows)} (workgroup, pair) samples; pair total median {np.median(total):.0f} cycles "
      f"(p10 {np.percentile(total, 10):.0f}, p90 {np.percentile(total, 90):.0f})")
for i, name in enumerate(NAMES):
    c = d[:, i]
    print(f"{i:2d} {name:32s} median {np.median(c):8.0f}  p10 {np.percentile(c, 10):8.0f}  p90 {np.percentile(c, 90):8.0f}"
          f"  {100 * np.median(c) / np.median(total):5.1f} %")
first = raw[:, 0, 0]
first = first[first != 0]
print(f"start spread of first pairs: {first.max() - first.min()} cycles")
