#!/usr/bin/env python
"""One short call that tells a fast box of the pool from a slow one and records why.

    python tools/box_class.py OUTDIR

1. tools/gpu_state.py --compact (driver's view + clock probe);
2. stage times of the 8-minute pair, cold, then again after 2 s of all-SIMD load (does the box only need waking?);
3. the driver's clock / power readings sampled at 20 Hz while ~400 masters run back to back (what the chip does
   under THIS workload, not under the probe).
"""
import json
import os
import statistics
import subprocess
import sys
import threading
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))


def main():
    out_dir = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/box"
    os.makedirs(out_dir, exist_ok=True)
    import ctypes

    import gpu_state
    import matchering_amd as mg
    from matchering_amd import _native
    from matchering_amd._native import STAGES
    from matchering_amd.device import Device
    from matchering_amd.synth import make_pair

    report = {"gpu_state": gpu_state.compact_state()}
    card = report["gpu_state"].get("card")
    dev = Device(0)
    native = mg.Config().to_native()
    target, reference = make_pair(480.0, 44100, pair=0)
    n, nr = target.shape[0], reference.shape[0]
    t_dev, r_dev, out = dev.upload(target), dev.upload(reference), dev.alloc(n * 8)
    dev.stage_timing(True)

    def stages(rounds):
        rows = []
        for _ in range(rounds):
            dev.master(t_dev, n, r_dev, nr, native, result=out, want_report=False)
            rows.append(dev.stage_times())
        return {s: round(statistics.median(r[s] for r in rows) * 1e3, 1) for s in STAGES if rows[0][s] is not None}

    stages(2)
    report["stage_us_cold"] = stages(7)
    import mgx_probe            # tools/mgx_probe.py (this folder is on sys.path: the script lives in it)

    res = [0.0] * 4
    t_end = time.perf_counter() + 2.0
    while time.perf_counter() < t_end:
        res = mgx_probe.clock(dev.index, 8192, 600000)
    report["preheat_shader_mhz"] = round(res[2], 1)
    report["stage_us_after_2s_of_load"] = stages(7)
    dev.stage_timing(False)

    samples, stop = [], threading.Event()

    def watch():
        while not stop.is_set():
            now = gpu_state.sysfs_state().get(card, {})
            samples.append((now.get("sclk") or {}).get("active"))
            samples.append(now.get("power_input_W", now.get("power_average_W")))
            time.sleep(0.05)

    th = threading.Thread(target=watch)
    th.start()
    t0 = time.perf_counter()
    steps = 0
    while time.perf_counter() - t0 < 1.5:
        for _ in range(20):
            dev.master(t_dev, n, r_dev, nr, native, result=out, want_report=False)
        dev.synchronize()
        steps += 20
    took = time.perf_counter() - t0
    stop.set()
    th.join()
    report["sustained"] = {"steps": steps, "ms_per_step": round(took / steps * 1e3, 4),
                           "sclk_seen": sorted({s for s in samples[0::2] if s}),
                           "power_W_seen": [min(p for p in samples[1::2] if p), max(p for p in samples[1::2] if p)]
                           if any(samples[1::2]) else None}
    sums = sum(report["stage_us_cold"].values())
    report["class"] = "slow" if report["stage_us_cold"].get("limit", 0) > 190 else "fast"
    report["stage_sum_us_cold"] = round(sums, 1)
    with open(os.path.join(out_dir, "box_class.json"), "w") as fh:
        json.dump(report, fh, indent=1)
    print(json.dumps(report, indent=1))
    if report["class"] == "slow":                   # a rare catch: keep everything the tools can say about it
        for name, cmd in (("rocm_smi_a.txt", ["rocm-smi", "-a"]), ("amd_smi_metric.txt", ["amd-smi", "metric"]),
                          ("amd_smi_static.txt", ["amd-smi", "static"])):
            try:
                text = subprocess.run(cmd, capture_output=True, text=True, timeout=60).stdout
                with open(os.path.join(out_dir, name), "w") as fh:
                    fh.write(text)
            except Exception:                       # noqa: BLE001
                pass


if __name__ == "__main__":
    main()
