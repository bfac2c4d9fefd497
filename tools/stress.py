"""Repeat mgx_master on one resident pair and compare every tenth result bit for bit with the first:
looks for intermittent faults and non-determinism (look-back, last-arriver decisions, tickets).

    python tools/stress.py SECONDS RATE FFT_SIZE ITERATIONS
"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import matchering_amd as mg
from matchering_amd.device import Device
from matchering_amd.synth import make_pair
from matchering_amd._native import check, library
seconds, rate, fft, iters = float(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
dev = Device(0)
cfg = mg.Config(internal_sample_rate=rate, fft_size=fft)
native = cfg.to_native()
t, r = make_pair(seconds, rate, pair=1)
n, nr = t.shape[0], r.shape[0]
td, rd, out, out2 = dev.upload(t), dev.upload(r), dev.alloc(n * 8), dev.alloc(n * 8)
dev.master(td, n, rd, nr, native, want_report=True, result=out)
ref = dev.download(out, (n, 2))
bad = 0
for i in range(iters):
    if i % 3 == 0:
        dev.master(td, n, rd, nr, native, want_report=False, result=None, result_no_limiter=out2)
    dev.master(td, n, rd, nr, native, want_report=(i % 7 == 0), result=out)
    if i % 10 == 9:
        dev.synchronize()
        got = dev.download(out, (n, 2))
        if not np.array_equal(got, ref):
            bad += 1
            print("iteration", i, "differs: max", float(np.abs(got - ref).max()), flush=True)
dev.synchronize()
print("stress", seconds, rate, fft, iters, "mismatches", bad, flush=True)
