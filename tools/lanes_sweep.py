"""Eight resident 4-minute pairs through 1, 2, 3, 4 and 8 device handles: GPU time per batch and the time the host
needs to submit it (python tools/lanes_sweep.py)."""
import sys, time
sys.path.insert(0, '/root/repo')
import numpy as np
import matchering_amd as mg
from matchering_amd.device import Device
from matchering_amd.synth import make_pair
cfg = mg.Config(); native = cfg.to_native()
pairs = [make_pair(240.0, 44100, pair=k) for k in range(8)]
for nl in (1, 2, 3, 4, 8):
    lanes = [Device(0) for _ in range(nl)]
    jobs = []
    for k, (t, r) in enumerate(pairs):
        d = lanes[k % nl]
        jobs.append((d, d.upload(t), t.shape[0], d.upload(r), r.shape[0], d.alloc(t.shape[0] * 8)))
    def step():
        for d, t, n, r, nr, out in jobs:
            d.master(t, n, r, nr, native, result=out, want_report=False)
    def sync():
        for d in lanes: d.synchronize()
    for _ in range(3): step()
    sync()
    best, submit = 1e9, 1e9
    for rep in range(3):
        t0 = time.perf_counter()
        for _ in range(5): step()
        t1 = time.perf_counter()
        sync()
        best = min(best, (time.perf_counter() - t0) / 5)
        submit = min(submit, (t1 - t0) / 5)
    print(f"{nl} lanes: {best*1e3:.3f} ms per 8 pairs (the host needs {submit*1e3:.3f} ms to submit them)")
    for j in jobs:
        for b in (j[1], j[3], j[5]): b.release()
    for d in lanes: d.close()
