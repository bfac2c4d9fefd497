#!/usr/bin/env python
"""bench.py -- stereo Msamples/s mastered on MI355X (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

One "step" = one pass of the whole hot path (``mgx_master`` = matchering
``stages.main``: level analysis of target and reference, FIR design, overlap-save
convolution, 4-round level correction, output) over one synthetic 8-minute
44.1 kHz stereo pair that is already resident in HBM.  Default workload =
BASELINE.json configs[1] ("single 8-minute pair, matching-EQ FIR only, limiter
bypassed" -> ``need_no_limiter`` output); configs[2] (full pipeline incl. the
Hyrax limiter) is timed next to it and reported under "full_pipeline".  With N
ranks every rank masters its own pair (pairs are independent: no data-path
collective, weak scaling); the FIR tables are all-gathered over RCCL after the
timed region, which is the only traffic that crosses xGMI.

torch is used for rendezvous/barrier/max-reduce only (gloo, CPU tensors); device
memory, streams and timing go through libmgx.  Prints ONE JSON line on rank 0.
"""

import argparse
import ctypes
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X spec (MI355X_MICROARCH.md); measured copy ceiling 6290
CONV_BYTES_PER_FRAME = 16      # read 8 + write 8 (the mid plane's 4 B are booked to S4)
MODEL_BYTES = {"8min_fir_only": 64, "8min_full": 72}     # SURVEY.md 8(d) whole-pipeline byte models


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default="8min_fir_only", choices=sorted(MODEL_BYTES))
    ap.add_argument("--seconds", type=float, default=480.0)
    ap.add_argument("--sample-rate", type=int, default=44100)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-secondary", action="store_true", help="skip the second workload and PCIe figure")
    return ap.parse_args()


class Ranks:
    """Rendezvous + barrier + max-reduce over the ranks torch.distributed.run started."""

    def __init__(self):
        self.rank = int(os.environ.get("RANK", "0"))
        self.world = int(os.environ.get("WORLD_SIZE", "1"))
        self.local = int(os.environ.get("LOCAL_RANK", "0"))
        self.dist = None
        if self.world > 1:
            import torch.distributed as dist

            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", "29511")
            dist.init_process_group("gloo", rank=self.rank, world_size=self.world)
            self.dist = dist

    def barrier(self):
        if self.dist:
            self.dist.barrier()

    def max(self, value):
        if not self.dist:
            return value
        import torch

        t = torch.tensor([value], dtype=torch.float64)
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)
        return float(t.item())

    def broadcast_bytes(self, payload, size):
        if not self.dist:
            return payload
        import torch

        t = torch.zeros(size, dtype=torch.uint8)
        if self.rank == 0:
            t[:] = torch.frombuffer(bytearray(payload), dtype=torch.uint8)
        self.dist.broadcast(t, src=0)
        return bytes(t.numpy().tobytes())

    def finish(self):
        if self.dist:
            self.dist.destroy_process_group()


def timed_steps(ranks, dev, step, steps, warmup):
    for _ in range(warmup):
        step()
    dev.synchronize()
    ranks.barrier()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    dev.synchronize()
    ranks.barrier()
    return ranks.max(time.perf_counter() - t0)


def main():
    args = parse()
    ranks = Ranks()
    if args.gpus != ranks.world:
        if ranks.world == 1 and args.gpus > 1:
            raise SystemExit("launch with torch.distributed.run --nproc-per-node N for --gpus N > 1")
    import matchering_amd as mg
    from matchering_amd._native import check, library
    from matchering_amd.device import Device
    from matchering_amd.synth import make_pair

    lib = library()
    from matchering_amd.device import device_count

    dev = Device(ranks.local % max(1, device_count()))      # one rank per GPU; wraps only when ranks outnumber GPUs
    cfg = mg.Config(internal_sample_rate=args.sample_rate)
    native = cfg.to_native()
    target, reference = make_pair(args.seconds, args.sample_rate, pair=ranks.rank)
    n, nr = target.shape[0], reference.shape[0]
    t_dev, r_dev = dev.upload(target), dev.upload(reference)
    out_a, out_b = dev.alloc(n * 8), dev.alloc(n * 8)

    def step_fir_only():
        dev.master(t_dev, n, r_dev, nr, native, result=None, result_no_limiter=out_a, want_report=False)

    def step_full():
        dev.master(t_dev, n, r_dev, nr, native, result=out_b, want_report=False)

    steps = {"8min_fir_only": step_fir_only, "8min_full": step_full}
    elapsed = timed_steps(ranks, dev, steps[args.workload], args.steps, args.warmup)
    frames_total = n * args.steps * ranks.world
    value = frames_total / elapsed / 1e6
    ms_per_step = elapsed / args.steps * 1e3

    line = {
        "metric": "stereo Msamples/s mastered (44.1 kHz pairs); % HBM roofline @1/2/4/8 GPU",
        "value": round(value, 2), "unit": "Msamples/s", "n_gpus": ranks.world, "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": round(ms_per_step, 4), "higher_is_better": True,
        "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": f"{args.workload}: {args.seconds:.0f} s stereo {args.sample_rate} Hz pair per GPU, "
                               f"fft_size {cfg.fft_size}, inputs resident in HBM",
                   "frames_per_pair": n, "pairs_per_step": ranks.world, "parallelism": f"pairs x{ranks.world}"},
        "pipeline_hbm_model": {"bytes_per_frame": MODEL_BYTES[args.workload],
                               "achieved_GBs": round(MODEL_BYTES[args.workload] * n / (elapsed / args.steps) / 1e9, 1),
                               "frac_of_8TBs": round(MODEL_BYTES[args.workload] * n / (elapsed / args.steps) / 1e9
                                                     / HBM_PEAK_GBS, 4)},
    }

    # ---- multi-GPU: all-gather the FIR tables over RCCL (off the timed path) ---------------
    if ranks.world > 1:
        # (never on the timed path; a failing collective is reported in the line, it does not lose the measurement)
        try:
            id_buf = ctypes.create_string_buffer(128)
            if ranks.rank == 0:
                check(lib.mgx_comm_unique_id(id_buf))
            uid = ranks.broadcast_bytes(id_buf.raw, 128)
            check(lib.mgx_comm_init(dev.handle, ctypes.c_char_p(uid), ranks.rank, ranks.world))
            taps_dev, taps = ctypes.c_void_p(), ctypes.c_int32()
            check(lib.mgx_last_fir(dev.handle, ctypes.byref(taps_dev), ctypes.byref(taps)))
            count = 2 * taps.value
            table = dev.alloc(count * 4 * ranks.world)
            check(lib.mgx_comm_allgather_f32(dev.handle, taps_dev, ctypes.c_void_p(table.ptr), count))
            dev.synchronize()
            firs = dev.download(table, (ranks.world, 2, taps.value))
            own = dev.download(int(taps_dev.value), (2, taps.value))
            ok = bool(np.array_equal(firs[ranks.rank], own)) and bool(np.all(np.isfinite(firs)))
            line["rccl_fir_allgather"] = {"bytes_per_rank": count * 4, "ok": ok}
            check(lib.mgx_comm_destroy(dev.handle))

        except Exception as exc:       # noqa: BLE001
            line["rccl_fir_allgather"] = {"ok": False, "error": str(exc)[:200]}

    if ranks.rank == 0:
        # ---- roofline of the dominant kernel (k_conv), HIP events on the kernel's stream ----
        rng = np.random.RandomState(0)
        f = cfg.fft_size
        hm = rng.randn(f) / np.sqrt(f)
        hs = rng.randn(f) / np.sqrt(f)
        dp = ctypes.POINTER(ctypes.c_double)
        ms = ctypes.c_float()
        mid_plane = dev.alloc(n * 4)
        for _ in range(2):
            check(lib.mgx_convolve_timed(dev.handle, ctypes.c_void_p(t_dev.ptr), n, hm.ctypes.data_as(dp),
                                         hs.ctypes.data_as(dp), f, 1.0, ctypes.c_void_p(out_a.ptr),
                                         ctypes.c_void_p(mid_plane.ptr), 20, ctypes.byref(ms)))
        achieved = CONV_BYTES_PER_FRAME * n / (ms.value * 1e-3) / 1e9
        # HBM bytes per launch from the PMC passes of tools/gpu_pmc.sh (FETCH_SIZE x2 on gfx950 +
        # WRITE_SIZE), valid for this workload only; rocprofv3 cannot wrap the process it runs in
        traffic = None
        pmc_path = os.path.join(ROOT, "profiles", "pmc_traffic.json")
        if os.path.exists(pmc_path) and args.seconds == 480.0 and args.sample_rate == 44100:
            with open(pmc_path) as fh:
                traffic = json.load(fh)["kernels"].get("k_conv<13>", {}).get("hbm_bytes_per_launch")
        line["roofline"] = {"kernel": "k_conv<13> (overlap-save FIR, B=8192)", "bound": "hbm",
                            "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                            "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic,
                            "kernel_ms": round(ms.value, 4),
                            "algorithmic_bytes_per_launch": CONV_BYTES_PER_FRAME * n}
        if not args.no_secondary and ranks.world == 1:
            other = "8min_full" if args.workload == "8min_fir_only" else "8min_fir_only"
            e2 = timed_steps(ranks, dev, steps[other], args.steps, 1)
            line["full_pipeline" if other == "8min_full" else "fir_only"] = {
                "workload": other, "value": round(n * args.steps / e2 / 1e6, 2), "unit": "Msamples/s",
                "ms_per_step": round(e2 / args.steps * 1e3, 4),
                "frac_of_8TBs": round(MODEL_BYTES[other] * n / (e2 / args.steps) / 1e9 / HBM_PEAK_GBS, 4)}
            # host buffers in, host buffers out (PCIe inclusive) -- never the headline value
            from matchering_amd import stages

            t0 = time.perf_counter()
            stages.main(target, reference, cfg, need_default=False, need_no_limiter=True, device=dev)
            line["pcie_inclusive"] = {"value": round(n / (time.perf_counter() - t0) / 1e6, 2), "unit": "Msamples/s",
                                      "note": "pageable numpy in/out through stages.main, one pair"}
            # config #4's share of one GPU: eight four-minute pairs, resident in HBM, submitted round-robin
            # to two device handles (two HIP streams) so that one pair's short serial kernels overlap
            # the other's streaming ones -- the batch front end's lanes (matchering_amd/batch.py)
            if args.seconds == 480.0:
                lanes = [dev, Device(ranks.local)]
                half_t, half_r = make_pair(args.seconds / 2, args.sample_rate, pair=100)
                nb, nrb = half_t.shape[0], half_r.shape[0]
                bufs = []
                for k in range(8):
                    d = lanes[k % 2]
                    bufs.append((d, d.upload(half_t), d.upload(half_r), d.alloc(nb * 8)))

                def batch_pass(only=None):
                    for d, tb, rb, ob in bufs:
                        (only or d).master(tb, nb, rb, nrb, native, result=ob, want_report=False)
                    for d in lanes:
                        d.synchronize()

                times = {}
                for name, only in (("one_lane", dev), ("two_lanes", None)):
                    batch_pass(only)
                    t0 = time.perf_counter()
                    for _ in range(3):
                        batch_pass(only)
                    times[name] = (time.perf_counter() - t0) / 3
                line["batch_8x4min_full"] = {
                    "value": round(8 * nb / times["two_lanes"] / 1e6, 2), "unit": "Msamples/s",
                    "ms_per_batch": round(times["two_lanes"] * 1e3, 3),
                    "one_lane_ms_per_batch": round(times["one_lane"] * 1e3, 3),
                    "note": "8 four-minute pairs per GPU (config #4's share), full pipeline, two handles"}
                for d, tb, rb, ob in bufs:
                    for b in (tb, rb, ob):
                        b.release()
                lanes[1].close()
        if not args.no_cpu_baseline and ranks.world == 1:
            sys.path.insert(0, os.path.join(ROOT, "oracle"))
            import mastering_oracle as mo

            ocfg = mo.params(internal_sample_rate=args.sample_rate)
            need = (False, True, False) if args.workload == "8min_fir_only" else (True, False, False)
            runs = []
            t_all = time.perf_counter()
            while len(runs) < 3 or (time.perf_counter() - t_all < 10.0 and len(runs) < 8):
                t0 = time.perf_counter()
                mo.master(target, reference, ocfg, *need)
                runs.append(time.perf_counter() - t0)
            cpu_s = min(runs)
            line["cpu_baseline"] = {"value": round(n / cpu_s / 1e6, 3), "unit": "Msamples/s", "cores": 1,
                                    "kind": "port", "seconds": round(cpu_s, 2), "runs": len(runs),
                                    "sample": f"the same {args.seconds:.0f} s pair, best of {len(runs)} runs of "
                                              f"oracle/mastering_oracle.py (numpy/scipy float64 restatement of "
                                              f"stages.main, single thread; {sum(runs):.0f} s of CPU work), "
                                              f"host has {os.cpu_count()} logical cores"}
            line["speedup_vs_cpu"] = round(value / line["cpu_baseline"]["value"], 1)
    if ranks.rank == 0:
        print(json.dumps(line))
    ranks.finish()


if __name__ == "__main__":
    main()
