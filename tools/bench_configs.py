"""Throughput of the other BASELINE.json configurations on one GPU (inputs resident in HBM): config #4's
four-minute 44.1 kHz pairs and config #5's 96 kHz pairs with a 16 k-tap matching FIR.  Not the bench line
(bench.py measures configs #2/#3); numbers go into DESIGN.md.

    python tools/bench_configs.py
"""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import matchering_amd as mg
from matchering_amd.device import Device
from matchering_amd.synth import make_pair


def run(name, seconds, rate, **cfg_kw):
    dev = Device(0)
    cfg = mg.Config(internal_sample_rate=rate, **cfg_kw)
    native = cfg.to_native()
    t, r = make_pair(seconds, rate, pair=1)
    n, nr = t.shape[0], r.shape[0]
    td, rd, out = dev.upload(t), dev.upload(r), dev.alloc(n * 8)
    res = {}
    for label, kw in (("fir_only", dict(result=None, result_no_limiter=out)), ("full", dict(result=out))):
        for _ in range(2):
            dev.master(td, n, rd, nr, native, want_report=False, **kw)
        dev.synchronize()
        t0 = time.perf_counter()
        for _ in range(5):
            dev.master(td, n, rd, nr, native, want_report=False, **kw)
        dev.synchronize()
        dt = (time.perf_counter() - t0) / 5
        res[label] = {"ms_per_pair": round(dt * 1e3, 3), "Msamples_per_s": round(n / dt / 1e6, 1)}
    print(json.dumps({"config": name, "frames": n, **res}))
    dev.close()


if __name__ == "__main__":
    run("4 min, 44.1 kHz, fft 4096 (config #4 pair)", 240.0, 44100)
    run("4 min, 96 kHz, fft 16384 (config #5 pair)", 240.0, 96000, fft_size=16384)
    run("15 min, 44.1 kHz, fft 4096 (maximum length)", 900.0, 44100)
