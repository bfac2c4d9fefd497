#!/usr/bin/env python
"""bench.py -- stereo Msamples/s mastered on MI355X (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W [--workload NAME]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

Both forms work for N > 1: started by a launcher (RANK / WORLD_SIZE / LOCAL_RANK in the environment) every
process is one rank; started plainly with ``--gpus N`` the script starts its N rank processes itself (ranks
wrap onto the visible GPUs when there are fewer than N) and rank 0 prints the line.

One "step" = one pass of the whole hot path (``mgx_master`` = matchering ``stages.main``: level
analysis of target and reference, FIR design, overlap-save convolution, 4-round level correction,
Hyrax limiter) over synthetic stereo pairs that are already resident in HBM.

Workloads (BASELINE.json configs):
  8min_full       configs[2]  one 8-minute 44.1 kHz pair, full pipeline incl. limiter -- the N=1 default,
                              the configuration the 40 %-of-roofline target is stated on
  8min_fir_only   configs[1]  the same pair, limiter bypassed (``result_no_limiter`` only)
  4min_x8_full    configs[3]  one GPU's share of "64 four-minute pairs over 8 GPUs": eight pairs per step,
                              submitted to three device handles (three HIP streams) -- the default for N > 1
  96k_16k_full    configs[4]  one 4-minute 96 kHz pair with a 16384-tap FIR (frequency-domain delay line)
  96k_16k_x16_full configs[4] one GPU's share of "batch 128 on 8 GPUs": sixteen such pairs per step through the lanes
With N ranks every rank masters its own pairs (pairs are independent: no data-path collective, weak
scaling); the FIR tables are all-gathered over RCCL after the timed region, the only traffic that
crosses xGMI.

Rendezvous, barrier and max-reduce go through matchering_amd.ranks (a local socket: no torch anywhere in this
file); device memory, streams and timing go through libmgx.  Prints ONE JSON line on rank 0.
"""

import argparse
import ctypes
import json
import os
import statistics
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X spec (MI355X_MICROARCH.md)
HBM_COPY_GBS = 6290.0          # measured float4 copy ceiling, same guide
# SURVEY.md 8(d) byte models, bytes per target frame
PIPELINE_BYTES = {"8min_full": 72, "8min_fir_only": 64, "4min_x8_full": 72, "96k_16k_full": 72, "96k_16k_x16_full": 72}
LANE_WORKLOADS = {"4min_x8_full": (44100, 4096, 240.0, 8), "96k_16k_x16_full": (96000, 16384, 240.0, 16)}
REPEAT_BLOCKS = 5              # further K-step blocks behind the one `value` is taken from (ms_per_step_repeats)
KERNEL_BYTES = {"convolve": 16,    # S3: read 8 + write 8 (the mid plane's 4 B are booked to S4)
                "limit": 16}       # read 8 + write 8 per launch of the one-pass limiter (S5's 24 B model
                                   # counts a second read that this kernel takes from the L2 / Infinity Cache)
WORKLOADS = sorted(PIPELINE_BYTES)


def kernel_name(stage, fft):
    """The kernel a stage's launch runs (csrc/mgx.hip run_conv / launch_limiter pick it from the FIR length)."""
    if stage == "limit":
        return "k_limit<256,4> (Hyrax limiter, one pass)"
    if fft == 4096:
        return "k_conv_wide<14> (overlap-save FIR, 4096 taps on 16384-point blocks)"
    if fft == 16384:
        return "k_conv_delay<14> (overlap-save FIR, two-partition frequency-domain delay line)"
    return "k_conv (overlap-save FIR, N = 2F)"


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default="auto", choices=["auto"] + WORKLOADS,
                    help="auto = 8min_full on one GPU, 4min_x8_full per rank on several (96k_16k_x16_full: config #5's "
                         "per-GPU share, sixteen 96 kHz pairs per rank)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-secondary", action="store_true", help="skip the side workloads and the PCIe figure")
    ap.add_argument("--no-traffic", action="store_true",
                    help="skip the two rocprofv3 counter passes that measure the dominant kernels' HBM traffic")
    ap.add_argument("--no-gpu-state", action="store_true", help="skip the clock / power / partition probe")
    ap.add_argument("--stand-in", action="store_true",
                    help="no GPU: a step is 1 ms of sleep (the multi-rank plumbing of this script, for the CPU tests)")
    ap.add_argument("--lanes", type=int, default=0,
                    help="device handles per GPU for 4min_x8_full; 0 = rank 0 measures two against three once and every "
                         "rank adopts its choice (batch.choose_lanes)")
    ap.add_argument("--spinup", type=float, default=0.5,
                    help="seconds of untimed steps before the warm-up (the device climbs out of its idle clocks)")
    return ap.parse_args()


from matchering_amd.ranks import Ranks  # noqa: E402  (rendezvous + barrier + max + broadcast over a local socket)


def _free_port():
    import socket

    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def launch_ranks(world, argv=None):
    """``python bench.py --gpus N`` without a launcher: start the N rank processes (this same script with RANK,
    LOCAL_RANK, WORLD_SIZE, MASTER_ADDR, MASTER_PORT set, as ``torch.distributed.run`` would), let rank 0's
    standard output through -- it prints the line -- and return the first non-zero exit code, 0 if none.  Rank r
    uses GPU r modulo the number of visible GPUs (no HIP_VISIBLE_DEVICES games: RCCL wants to see the peers)."""
    argv = list(sys.argv[1:] if argv is None else argv)
    port = _free_port()                       # only a name for the rendezvous socket (matchering_amd.ranks)
    procs = []
    for rank in range(world):
        env = dict(os.environ, RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), MGX_BENCH_SELF_LAUNCHED="1")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__), *argv], env=env,
                                      stdout=None if rank == 0 else subprocess.DEVNULL))
    code = 0
    try:
        pending = list(procs)
        while pending:
            for p in list(pending):
                rc = p.poll()
                if rc is None:
                    continue
                pending.remove(p)
                if rc != 0 and code == 0:
                    code = rc
                    for q in pending:           # a rank that died takes the job with it: the others would wait for it
                        q.terminate()
            time.sleep(0.05)
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
    return code


def spin_up(sync, step, seconds):
    """Untimed steps for ``seconds`` before the warm-up.  The device sleeps while the host builds the synthetic
    pairs (sclk ~100 MHz, deep sleep enabled) and W = 3-5 warm-up steps are only 2 ms; a mastering service is a
    busy device, so that is the state the K timed steps should see.  On the boxes measured it makes no
    difference (0.504 ms per step with and without; profiles/r03_e_*, r03_i_*): it is insurance against a box
    that climbs out of idle slowly, not a speed-up.  Returns the steps run."""
    done, t0 = 0, time.perf_counter()
    spin_up.recovered = 0
    while time.perf_counter() - t0 < seconds:
        try:
            for _ in range(8):
                step()
            sync()
        except Exception as exc:
            # a bounded device wait expired and the handle has switched to its safe mode ("call again", DESIGN.md
            # section 1): this loop is rank-local and untimed, so it is the place to absorb that -- the timed steps
            # then run in the mode the handle has settled in, and the line says so (spinup.recovered_calls)
            if "call again" not in str(exc) or spin_up.recovered >= 8:
                raise
            spin_up.recovered += 1
            continue
        done += 8
    return done


def timed_steps(ranks, sync, step, steps, warmup, own=None):
    """W untimed steps, then K steps between two barriers (each behind a stream synchronisation); returns the
    maximum over ranks.  ``own`` (a list) receives this rank's time up to its own synchronisation, before the
    closing barrier: the spread over ranks tells a straggler from a uniformly slow job."""
    for _ in range(warmup):
        step()
    sync()
    ranks.barrier()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    sync()
    mine = time.perf_counter() - t0
    ranks.barrier()
    if own is not None:
        own.append(mine)
    return ranks.max(time.perf_counter() - t0)


class Workload:
    """Resident inputs/outputs of one named workload on this rank and the function that runs one step."""

    def __init__(self, name, rank, local, mg, Device, device_count, make_pair, world=1, lanes=0):
        self.name = name
        from matchering_amd.batch import choose_lanes, lane_choice_report, lane_device

        index = local % max(1, device_count())          # one rank per GPU; wraps only when ranks outnumber GPUs
        self.dev = lane_device(index, 0)                 # (the process-wide handles of matchering_amd.batch)
        self.lanes = [self.dev]
        self.lane_choice = None
        self.sample_rate, self.fft, self.seconds, self.pairs = 44100, 4096, 480.0, 1
        if name in LANE_WORKLOADS:
            self.sample_rate, self.fft, self.seconds, self.pairs = LANE_WORKLOADS[name]
            # two or three device handles, whichever a short measured batch says is faster on THIS GPU
            # (batch.choose_lanes: a pair's latency-bound stretches -- FIR design, level decisions, the
            # limiter's look-back waits -- are filled by the other pairs' kernels; boxes of the pool disagree
            # on whether the third handle still pays).  Outside the timed region.
            from matchering_amd.batch import lanes_allowed

            if lanes > 0:                                       # given (or decided by rank 0): nothing is measured here
                count = min(lanes, lanes_allowed(world))
                self.lane_choice = {"lanes": count, "given": True}
            else:
                count = choose_lanes(index, world_size=world)  # (ranks that share a GPU share its lane budget)
                self.lane_choice = lane_choice_report(index)
            self.lanes = [lane_device(index, k) for k in range(count)]
        elif name == "96k_16k_full":
            self.sample_rate, self.fft, self.seconds = 96000, 16384, 240.0
        self.cfg = mg.Config(internal_sample_rate=self.sample_rate, fft_size=self.fft)
        self.native = self.cfg.to_native()
        self.want_limiter = name != "8min_fir_only"
        self.jobs = []
        self.host_pair = None
        # (sixteen 96 kHz pairs are a minute of host synthesis: four are synthesised, the other twelve are those with
        # the channels swapped and / or the target a little quieter -- distinct audio, distinct level decisions and
        # FIRs, the same kernels and bytes)
        self.synthesised = min(self.pairs, 4) if name == "96k_16k_x16_full" else self.pairs
        base = []
        for k in range(self.pairs):
            if k < self.synthesised:
                target, reference = make_pair(self.seconds, self.sample_rate, pair=rank * self.pairs + k)
                base.append((target, reference))
            else:
                bt, br = base[k % self.synthesised]
                variant = k // self.synthesised
                target = np.ascontiguousarray(bt[:, ::-1] if variant & 1 else bt) * np.float32(1.0 - 0.03 * variant)
                reference = np.ascontiguousarray(br[:, ::-1]) if variant & 2 else br
            if k == 0:
                self.host_pair = (target, reference)
            d = self.lanes[k % len(self.lanes)]
            self.jobs.append((d, d.upload(target), target.shape[0], d.upload(reference), reference.shape[0],
                              d.alloc(target.shape[0] * 8)))
        self.frames = sum(j[2] for j in self.jobs)

    def step(self):
        for d, t, n, r, nr, out in self.jobs:
            if self.want_limiter:
                d.master(t, n, r, nr, self.native, result=out, want_report=False)
            else:
                d.master(t, n, r, nr, self.native, result=None, result_no_limiter=out, want_report=False)

    def sync(self):
        for d in self.lanes:
            d.synchronize()

    def describe(self):
        if self.name in LANE_WORKLOADS:
            which = "#4" if self.name == "4min_x8_full" else "#5"
            what = (f"{self.pairs} x {self.seconds:.0f} s stereo {self.sample_rate} Hz pairs per GPU (config {which}'s per-GPU "
                    f"share), {len(self.lanes)} device handles (measured choice)")
            if self.synthesised < self.pairs:
                what += (f"; {self.synthesised} pairs synthesised, the others derived from them (channels swapped, target "
                         f"scaled)")
        else:
            what = f"{self.seconds:.0f} s stereo {self.sample_rate} Hz pair per GPU"
        tail = "full pipeline incl. Hyrax limiter" if self.want_limiter else "matching-EQ FIR only (limiter bypassed)"
        return f"{self.name}: {what}, fft_size {self.fft}, {tail}, inputs resident in HBM"

    def stage_profile(self, steps):
        """Device time per stage (HIP events on the handle's stream, inside mgx_master), median over steps."""
        d, t, n, r, nr, out = self.jobs[0]
        d.stage_timing(True)
        rows = []
        for _ in range(steps):
            if self.want_limiter:
                d.master(t, n, r, nr, self.native, result=out, want_report=False)
            else:
                d.master(t, n, r, nr, self.native, result=None, result_no_limiter=out, want_report=False)
            rows.append(d.stage_times())
        d.stage_timing(False)
        return {k: statistics.median(row[k] for row in rows) for k in rows[0] if rows[0][k] is not None}, n

    def release(self):
        for d, t, n, r, nr, out in self.jobs:
            for b in (t, r, out):
                b.release()


def measure_traffic(workload, timeout=150):
    """HBM bytes per launch of the streaming kernels from the L2's memory-side counters, measured NOW: one
    rocprofv3 pass per counter (FETCH_SIZE and WRITE_SIZE do not fit one pass) over a short run of this same
    script (rocprofv3 cannot wrap the process it runs in).  MI355X_MICROARCH.md, HBM: FETCH_SIZE is in KiB and
    counts 64 B per 128-B request on gfx950 (x2); WRITE_SIZE is in KiB, uncorrected (calibrated in
    profiles/pmc_traffic.json on kernels of known byte counts).  Returns {stage: bytes} or {} when the
    profiler is not usable here."""
    import csv
    import shutil
    import subprocess
    import tempfile
    from collections import defaultdict

    if not shutil.which("rocprofv3"):
        return {}
    if any(k.startswith(("ROCPROF", "ROCP_", "ROCPROFILER_")) for k in os.environ):
        return {}                                          # already running under a profiler: no nesting
    kib, last_step = {}, {}
    for counter in ("FETCH_SIZE", "WRITE_SIZE"):
        folder = tempfile.mkdtemp(prefix="mgx_pmc_", dir="/tmp")
        try:
            cmd = ["rocprofv3", "--pmc", counter, "--kernel-trace", "--output-format", "csv", "-d", folder, "-o", "r",
                   "--", sys.executable, os.path.abspath(__file__), "--steps", "2", "--warmup", "1", "--spinup", "0",
                   "--no-cpu-baseline", "--no-secondary", "--no-traffic", "--no-gpu-state", "--workload", workload]
            subprocess.run(cmd, env=dict(os.environ, TMPDIR="/tmp"), cwd="/tmp", stdout=subprocess.DEVNULL,
                           stderr=subprocess.DEVNULL, timeout=timeout, check=True)
            acc = defaultdict(list)
            rows = []
            for root, _, files in os.walk(folder):
                for name in files:
                    if name.endswith("counter_collection.csv"):
                        with open(os.path.join(root, name), newline="") as fh:
                            for row in csv.DictReader(fh):
                                if row.get("Counter_Name") == counter:
                                    acc[row["Kernel_Name"]].append(float(row["Counter_Value"]))
                                    rows.append((int(row.get("Dispatch_Id") or len(rows)), row["Kernel_Name"],
                                                 float(row["Counter_Value"])))
            kib[counter] = dict(acc)
            # one whole step: every dispatch from the last k_analyze on (the plan's one-off kernels ran long before)
            rows.sort()
            starts = [i for i, r in enumerate(rows) if "k_analyze" in r[1]]
            if starts:
                per = defaultdict(float)
                for _, kernel, value in rows[starts[-1]:]:
                    per[kernel] += value
                last_step[counter] = dict(per)
        except Exception:                                   # noqa: BLE001 -- a measurement aid: never lose the line
            return {}
        finally:
            shutil.rmtree(folder, ignore_errors=True)
    out = {}
    # (k_conv<..>, k_conv_wide<..>, k_conv_delay<..>: whichever the FIR length selects; not their *_prep kernels)
    for stage, wanted in (("convolve", lambda k: "k_conv" in k and "prep" not in k and "direct" not in k),
                          ("limit", lambda k: "k_limit" in k)):
        fetch = [max(v) for k, v in kib["FETCH_SIZE"].items() if wanted(k)]
        write = [max(v) for k, v in kib["WRITE_SIZE"].items() if wanted(k)]
        if fetch and write:
            out[stage] = int(max(fetch) * 1024 * 2 + max(write) * 1024)
    if len(last_step) == 2:
        by_kernel = {k: int((last_step["FETCH_SIZE"].get(k, 0.0) * 2 + last_step["WRITE_SIZE"].get(k, 0.0)) * 1024)
                     for k in set(last_step["FETCH_SIZE"]) | set(last_step["WRITE_SIZE"])}
        out["step_bytes"] = sum(by_kernel.values())
        out["step_kernels"] = {k.split("(")[0][:48]: v for k, v in sorted(by_kernel.items(), key=lambda kv: -kv[1]) if v > 1e6}
    return out


def roofline_of(kernel, ms, frames, traffic, fft=4096):
    alg = KERNEL_BYTES[kernel] * frames
    achieved = alg / (ms * 1e-3) / 1e9
    return {"kernel": kernel_name(kernel, fft), "bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS,
            "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 4),
            "frac_of_measured_copy_ceiling": round(achieved / HBM_COPY_GBS, 4), "traffic": traffic,
            "kernel_ms": round(ms, 4), "algorithmic_bytes_per_launch": alg,
            "timing": "HIP events around the launch inside mgx_master, on the stream it runs on, median of the "
                      "profiled steps (cold inputs: the whole pipeline runs between two launches)"}


class StandIn:
    """A workload without a GPU: ``--stand-in`` (tests/test_bench_launch.py drives the N > 1 path of this script --
    self-launch, rendezvous, barriers, max over ranks, the line -- on a machine that has none).  A step sleeps."""

    name, frames, pairs, lane_choice = "stand_in", 44100, 1, None

    def __init__(self, stands_for="auto"):
        self.stands_for = stands_for
        self.fft = 16384 if stands_for.startswith("96k_16k") else 4096
        if stands_for in LANE_WORKLOADS:
            rate, _, seconds, self.pairs = LANE_WORKLOADS[stands_for]
            self.frames = int(rate * seconds) * self.pairs

    def step(self):
        time.sleep(0.001)

    def sync(self):
        pass

    def describe(self):
        return (f"stand-in for {self.stands_for}: no GPU, one step = 1 ms of sleep (plumbing test of the multi-rank path; "
                f"frames and pairs per step are the workload's)")


def share_one_gpu_over_rccl(rank):
    """More ranks than GPUs (a 1-GPU box running ``--gpus 2``): RCCL refuses two ranks of one host on one device
    ("Duplicate GPU detected").  Giving every rank its own NCCL_HOSTID makes them look like one-GPU hosts that
    talk over the loop-back socket transport -- enough to prove the exchange end to end where no second GPU
    exists; never set when every rank has its own GPU."""
    os.environ["NCCL_HOSTID"] = f"mgx-bench-rank-{rank}"
    # ... and the limiters of two processes on one chip must not wait for each other (batch.ranks_share_a_gpu)
    os.environ.setdefault("MGX_LIMIT_TICKETS", "1")
    os.environ.setdefault("NCCL_SOCKET_IFNAME", "lo")
    os.environ.setdefault("NCCL_IB_DISABLE", "1")
    os.environ.setdefault("NCCL_SHM_DISABLE", "1")
    os.environ.setdefault("NCCL_P2P_DISABLE", "1")


def main():
    args = parse()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        raise SystemExit(launch_ranks(args.gpus))                  # this process only starts the ranks and waits
    ranks = Ranks()
    if args.gpus != ranks.world:
        raise SystemExit(f"--gpus {args.gpus} but the launcher started {ranks.world} ranks")
    try:
        run(args, ranks)
    finally:
        ranks.finish()


def run(args, ranks):
    # ONE line on standard output, and nothing else: RCCL prints a version banner there when a communicator is
    # created (through C stdio, flushed at exit -- i.e. AFTER the line).  The line goes to a private copy of the
    # descriptor; descriptor 1 itself is pointed at standard error for the rest of the process.
    sys.stdout.flush()
    line_out = os.fdopen(os.dup(1), "w")
    os.dup2(2, 1)
    line = {
        "metric": "stereo Msamples/s mastered (44.1 kHz pairs); % HBM roofline @1/2/4/8 GPU",
        "unit": "Msamples/s", "n_gpus": ranks.world, "steps": args.steps, "warmup": args.warmup,
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "launch": "self" if os.environ.get("MGX_BENCH_SELF_LAUNCHED") else ("launcher" if ranks.world > 1 else "single"),
    }
    if args.stand_in:
        wl, lib, name, sharing = StandIn(args.workload), None, "stand_in", False
    else:
        import matchering_amd as mg
        from matchering_amd._native import check, library
        from matchering_amd.device import Device, device_count
        from matchering_amd.synth import make_pair

        sharing = ranks.world > max(1, device_count())
        if sharing:
            share_one_gpu_over_rccl(ranks.rank)
        lib = library()
        name = args.workload
        if name == "auto":
            name = "8min_full" if ranks.world == 1 else "4min_x8_full"
        lanes = args.lanes
        if name in LANE_WORKLOADS and lanes <= 0 and ranks.world > 1:
            # ONE calibration per job, not one per rank: rank 0 measures two handles against three on its GPU (the GPUs
            # of a node are one model) while the others wait at a barrier, and every rank adopts the choice
            from matchering_amd.batch import choose_lanes, lane_choice_report

            mine = None
            if ranks.rank == 0:
                count = choose_lanes(ranks.local % max(1, device_count()), world_size=ranks.world)
                mine = dict(lane_choice_report(ranks.local % max(1, device_count())) or {"lanes": count}, decided_by="rank 0")
            decided = ranks.gather(mine)[0]
            lanes = int(decided["lanes"])
        wl = Workload(name, ranks.rank, ranks.local, mg, Device, device_count, make_pair, world=ranks.world, lanes=lanes)
        if name in LANE_WORKLOADS and args.lanes <= 0 and ranks.world > 1:
            wl.lane_choice = decided
    spun = spin_up(wl.sync, wl.step, args.spinup)
    own_seconds = []
    elapsed = timed_steps(ranks, wl.sync, wl.step, args.steps, args.warmup, own_seconds)
    # REPEAT_BLOCKS further blocks of the same K steps, each between its own two barriers: `value` stays the FIRST block
    # (so that `steps` says what it was measured on), the spread of all of them stands beside it
    blocks_ms = [elapsed / args.steps * 1e3]
    for _ in range(REPEAT_BLOCKS):
        blocks_ms.append(timed_steps(ranks, wl.sync, wl.step, args.steps, 0) / args.steps * 1e3)
    frames_total = wl.frames * args.steps * ranks.world
    value = frames_total / elapsed / 1e6
    ms_per_step = elapsed / args.steps * 1e3
    model = PIPELINE_BYTES.get(name, 0)
    pipeline_gbs = model * wl.frames * ranks.world / (elapsed / args.steps) / 1e9
    # which physical GPU every rank sits on (PCI address from the HIP runtime): "N ranks on N GPUs" is checkable, and
    # the roofline is taken against the GPUs that really exist -- ranks that share one share its 8 TB/s
    if args.stand_in:
        my_gpu = f"stand-in-{ranks.rank}"
    else:
        from matchering_amd.device import pci_bus_id

        my_gpu = pci_bus_id(ranks.local % max(1, device_count()))
    rank_gpus = ranks.gather(my_gpu)
    physical = len(set(rank_gpus))

    line.update({
        "value": round(value, 2), "ms_per_step": round(ms_per_step, 4),
        "ms_per_step_repeats": {"min": round(min(blocks_ms), 4), "median": round(statistics.median(blocks_ms), 4),
                                "max": round(max(blocks_ms), 4), "blocks": len(blocks_ms),
                                "note": f"{len(blocks_ms)} blocks of {args.steps} steps each, back to back, each between two "
                                        "barriers (maximum over ranks); the first one is `ms_per_step` / `value`"},
        "spinup": {"seconds": args.spinup, "steps": spun, "recovered_calls": getattr(spin_up, "recovered", 0),
                   "note": "untimed steps before the W warm-up steps, so that the K timed steps see a busy device rather "
                           "than the climb out of idle; no measurable effect on the boxes seen so far"},
        "config": {"workload": wl.describe(), "frames_per_gpu_per_step": wl.frames,
                   "pairs_per_gpu_per_step": wl.pairs, "parallelism": f"pairs x{ranks.world * wl.pairs}",
                   **({"lane_choice": wl.lane_choice} if wl.lane_choice else {})},
        "rank_gpus": rank_gpus, "physical_gpus": physical, "shared_gpu": physical < ranks.world,
        # how k_limit hands out its chunks: by workgroup number, or by atomic ticket where rank
        # PROCESSES share a chip (two limiter launches resident together must not wait for each other)
        "limiter_chunks": ("atomic ticket" if os.environ.get("MGX_LIMIT_TICKETS") == "1"
                           else "atomic ticket on the handles that recovered from an expired wait during the spin-up, workgroup "
                                "number on the others" if getattr(spin_up, "recovered", 0) else "workgroup number"),
        "pipeline_hbm_model": {"bytes_per_frame": model, "achieved_GBs": round(pipeline_gbs, 1),
                               "peak_GBs": HBM_PEAK_GBS * physical,
                               "frac_of_8TBs": round(pipeline_gbs / (HBM_PEAK_GBS * physical), 4),
                               "frac_of_6p29TBs": round(pipeline_gbs / (HBM_COPY_GBS * physical), 4),
                               "note": "whole job: bytes of all ranks against the one-GPU figures x the PHYSICAL GPUs the "
                                       "ranks sit on (rank_gpus); measured_bytes = HBM bytes per step by the PMC counters, "
                                       "all kernels (null when the profiler cannot be used)",
                               "measured_bytes": None, "measured_frac_of_8TBs": None},
    })
    # every rank's own time for the K steps (up to its own stream synchronisation, before the closing barrier)
    line["rank_seconds"] = [round(v, 6) for v in ranks.gather(own_seconds[0])]

    # ---- multi-GPU: all-gather the FIR tables over RCCL (off the timed path) ---------------
    # A failing collective is reported in the line, it does not lose the measurement -- and neither does one
    # that never returns: the exchange runs in a thread the main one waits for two minutes at most.
    exchange = {}

    def exchange_fir_tables():
        try:
            dev = wl.dev
            id_buf = ctypes.create_string_buffer(128)
            if ranks.rank == 0:
                check(lib.mgx_comm_unique_id(id_buf))
            uid = ranks.broadcast_bytes(id_buf.raw, 128)
            t_init = time.perf_counter()
            check(lib.mgx_comm_init(dev.handle, ctypes.c_char_p(uid), ranks.rank, ranks.world))
            exchange["init_s"] = time.perf_counter() - t_init
            seen = ctypes.c_int32()
            check(lib.mgx_comm_count(dev.handle, ctypes.byref(seen)))
            taps_dev, taps = ctypes.c_void_p(), ctypes.c_int32()
            check(lib.mgx_last_fir(dev.handle, ctypes.byref(taps_dev), ctypes.byref(taps)))
            count = 2 * taps.value
            table = dev.alloc(count * 4 * ranks.world)
            check(lib.mgx_comm_allgather_f32(dev.handle, taps_dev, ctypes.c_void_p(table.ptr), count))
            dev.synchronize()
            firs = dev.download(table, (ranks.world, 2, taps.value))
            own = dev.download(int(taps_dev.value), (2, taps.value))
            ok = bool(np.array_equal(firs[ranks.rank], own)) and bool(np.all(np.isfinite(firs)))
            # every rank mastered its own pairs, so the gathered tables must differ from rank to rank
            distinct = len({firs[r].tobytes() for r in range(ranks.world)})
            exchange["result"] = {"bytes_per_rank": count * 4, "ok": ok, "ranks_seen": int(seen.value),
                                  "rccl_init_s": round(exchange.get("init_s", 0.0), 3),
                                  "distinct_tables": distinct,
                                  "transport": ("loop-back sockets (ranks share a GPU: NCCL_HOSTID made distinct per rank)"
                                                if sharing else "RCCL default (xGMI peer-to-peer between the node's GPUs)")}
            check(lib.mgx_comm_destroy(dev.handle))
        except Exception as exc:       # noqa: BLE001
            exchange["result"] = {"ok": False, "error": str(exc)[:200]}

    stuck = False
    if ranks.world > 1 and not args.stand_in:
        import threading

        worker = threading.Thread(target=exchange_fir_tables, daemon=True)
        worker.start()
        worker.join(120.0)
        stuck = worker.is_alive()
        mine = ({"ok": False, "error": "no answer within 120 s"} if stuck
                else exchange.get("result", {"ok": False, "error": "no result"}))
        # (every rank gets here within its 120 s, stuck or not, so this gather cannot wait for a missing one)
        everyone = ranks.gather(mine)                          # the line carries rank 0's view and how many ranks agree
        line["rccl_fir_allgather"] = dict(everyone[0], ranks_ok=sum(1 for e in everyone if e.get("ok")))
        # the slowest rank's ncclCommInitRank: a box on which RCCL takes minutes to come up shows HERE, not in a timeout
        line["rccl_init_s"] = max((e.get("rccl_init_s") or 0.0) for e in everyone)
    elif ranks.world > 1:
        line["rendezvous"] = {"ranks_seen": len(ranks.gather(ranks.rank)), "transport": "matchering_amd.ranks"}
        # the exchange's shape without RCCL: every rank's (fake) FIR pair, 2 x fft_size float32, gathered over the
        # rendezvous socket -- what the GPU path all-gathers over xGMI
        count = 2 * wl.fft
        t_init = time.perf_counter()
        tables = ranks.gather(np.full(count, float(ranks.rank + 1), np.float32).tobytes())
        line["rccl_fir_allgather"] = {"bytes_per_rank": count * 4, "ok": len(tables) == ranks.world and
                                      all(len(t) == count * 4 for t in tables), "ranks_seen": len(tables),
                                      "distinct_tables": len(set(tables)), "rccl_init_s": None,
                                      "transport": "matchering_amd.ranks (stand-in: no RCCL without a GPU)",
                                      "ranks_ok": len(tables)}
        line["rccl_init_s"] = None

    if ranks.world > 1 and not stuck:
        # the SAME workload on one rank alone (the other ranks are idle, their GPUs too): the N = 1 point of a scaling
        # curve that is one workload -- the N = 1 default of this script is the 8-minute pair, another workload
        alone = Ranks(rank=0, world=1, local=ranks.local)
        if ranks.rank == 0:
            e1 = timed_steps(alone, wl.sync, wl.step, args.steps, 1)
            line["n1_same_workload"] = {"value": round(wl.frames * args.steps / e1 / 1e6, 2), "unit": "Msamples/s",
                                        "ms_per_step": round(e1 / args.steps * 1e3, 4), "workload": name,
                                        "note": "rank 0 alone right after the timed region, same resident pairs and lanes; "
                                                "value / (n_gpus x this) is the weak-scaling efficiency of ONE workload"}
        ranks.barrier()                              # (the others must not start tearing down under rank 0's run)
    if ranks.rank == 0 and not args.stand_in and not stuck:
        # ---- rooflines of the two streaming kernels, timed where they run: inside the pipeline ----
        # (at N > 1 the other ranks are idle by now: these legs describe one GPU, as at N = 1)
        stage_ms, n0 = wl.stage_profile(max(5, min(args.steps, 20)))
        line["stage_ms"] = {k: round(v, 4) for k, v in stage_ms.items()}
        # HBM traffic by the PMC counters, measured in this run (two rocprofv3 passes over a short child run of
        # the same workload); null when the profiler cannot be used
        traffic = {} if args.no_traffic else measure_traffic(name)
        kernels = [k for k in ("convolve", "limit") if k in stage_ms]
        per = {k: roofline_of(k, stage_ms[k], n0, traffic.get(k), wl.fft) for k in kernels}
        if traffic.get("step_bytes"):
            pm = line["pipeline_hbm_model"]
            pm["measured_bytes"] = traffic["step_bytes"] * wl.pairs
            pm["measured_frac_of_8TBs"] = round(traffic["step_bytes"] * wl.pairs / (ms_per_step * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)
            pm["measured_bytes_by_kernel"] = traffic.get("step_kernels")
        for k in kernels:
            per[k]["traffic_source"] = ("rocprofv3 --pmc FETCH_SIZE (x2, KiB) + --pmc WRITE_SIZE (KiB), one pass each, "
                                        "child runs of this script in this call" if k in traffic else None)
        dominant = max(kernels, key=lambda k: stage_ms[k])
        line["roofline"] = per[dominant]
        line["roofline_other"] = [per[k] for k in kernels if k != dominant]
        if not args.no_gpu_state:
            try:
                sys.path.insert(0, os.path.join(ROOT, "tools"))
                from gpu_state import compact_state

                line["gpu_state"] = compact_state(wl.dev)
            except Exception as exc:                        # noqa: BLE001
                line["gpu_state"] = {"error": repr(exc)[:200]}

        if not args.no_secondary and ranks.world == 1:
            alone = Ranks(rank=0, world=1, local=ranks.local)
            side = {}
            for other in WORKLOADS:
                if other == name:
                    continue
                w2 = Workload(other, 0, ranks.local, mg, Device, device_count, make_pair)
                spin_up(w2.sync, w2.step, min(args.spinup, 0.2))
                e2 = timed_steps(alone, w2.sync, w2.step, max(3, args.steps // 2), 1)
                per_step = e2 / max(3, args.steps // 2)
                side[other] = {"value": round(w2.frames / per_step / 1e6, 2), "unit": "Msamples/s",
                               "ms_per_step": round(per_step * 1e3, 4),
                               "frac_of_8TBs": round(PIPELINE_BYTES[other] * w2.frames / per_step / 1e9 / HBM_PEAK_GBS, 4),
                               "workload": w2.describe(), **({"lane_choice": w2.lane_choice} if w2.lane_choice else {})}
                w2.release()
            line["other_workloads"] = side
            # host buffers in, host buffers out (PCIe inclusive) -- never the headline value
            from matchering_amd import stages

            target, reference = wl.host_pair
            took = []
            for _ in range(3):             # the first call fills the pinned-host and HBM block pools
                t0 = time.perf_counter()
                stages.main(target, reference, wl.cfg, need_default=wl.want_limiter,
                            need_no_limiter=not wl.want_limiter, device=wl.dev)
                took.append(time.perf_counter() - t0)
            line["pcie_inclusive"] = {"value": round(target.shape[0] / min(took) / 1e6, 2), "unit": "Msamples/s",
                                      "note": "pageable numpy in -> pinned numpy out through stages.main, one pair, "
                                              "best of 3 calls"}
            line["pcie_inclusive_batch"] = host_to_host_batch(mg, make_pair)
            try:
                line["file_to_file"] = file_to_file(mg, wl.host_pair)
            except Exception as exc:        # noqa: BLE001 -- an unwritable temp folder must not lose the measurement
                line["file_to_file"] = {"error": repr(exc)}
            try:
                line["first_call"] = first_call(wl.host_pair)
                line["first_call_ms"] = line["first_call"].get("first_call_ms")
            except Exception as exc:        # noqa: BLE001
                line["first_call"] = {"error": repr(exc)[:200]}
        if not args.no_cpu_baseline:
            # the timed workload's own output (first pair, as the last timed step left it in HBM), fetched before
            # anything else runs on the handle, against what the oracle computes from the same float32 inputs
            d, t, n, r, nr, out = wl.jobs[0]
            got = d.download(out, (n, 2)).astype(np.float64)
            line["cpu_baseline"], oracle_out = cpu_baseline(wl, name, all_cores=ranks.world == 1)
            # (per GPU: the baseline is one host process against one GPU's share of the job)
            line["speedup_vs_cpu"] = round(value / ranks.world / line["cpu_baseline"]["value"], 1)
            diff = got - oracle_out
            line["parity"] = {"rms": float(np.sqrt(np.mean(diff * diff))), "max_abs": float(np.abs(diff).max()),
                              "tolerance_rms": 1e-5, "ok": bool(np.sqrt(np.mean(diff * diff)) <= 1e-5),
                              "against": "oracle/mastering_oracle.py (float64) on the workload's first pair; the "
                                         "GPU result is the one the last timed step left in HBM"}
    if ranks.rank == 0:
        print(json.dumps(line), file=line_out, flush=True)
    if stuck:                       # (a rank still inside the collective would keep the others' teardown waiting)
        os._exit(0)


FIRST_CALL_CHILD = r"""
import time
t_start = time.perf_counter()
import json, sys
import numpy as np
sys.path.insert(0, {root!r})
target, reference = np.load({tp!r}), np.load({rp!r})
t_loaded = time.perf_counter()
import matchering_amd as mg
from matchering_amd import stages
t_imported = time.perf_counter()
cfg = mg.Config()
stages.main(target, reference, cfg)
t_first = time.perf_counter()
stages.main(target, reference, cfg)
t_second = time.perf_counter()
other = mg.Config(lowess_frac=0.04)          # another Config: its raw -> smooth operator is built anew, all else is warm
stages.main(target, reference, other)
t_third = time.perf_counter()
print("FIRST " + json.dumps({{"import_ms": (t_imported - t_loaded) * 1e3, "first_main_ms": (t_first - t_imported) * 1e3,
                             "second_main_ms": (t_second - t_first) * 1e3, "third_main_new_config_ms": (t_third - t_second) * 1e3}}))
"""


def first_call(pair):
    """What the FIRST mastering of a fresh process costs (missing #5): the reference pays its whole cost
    in every process() call (core.py:32-121); here the first call of a process also loads the code object, creates the
    handle, fills the twiddle tables and the block pools and builds the Config's raw -> smooth operator -- none of which
    is inside any other number of this line.  A child process: import -> first stages.main result on the workload's pair
    (host arrays in, host arrays out), then the same call again (warm), then a call with another Config (only the
    operator is new)."""
    import shutil
    import tempfile

    folder = tempfile.mkdtemp(prefix="mgx_first_", dir="/dev/shm" if os.path.isdir("/dev/shm") else None)
    try:
        tp, rp = os.path.join(folder, "t.npy"), os.path.join(folder, "r.npy")
        np.save(tp, pair[0])
        np.save(rp, pair[1])
        run = subprocess.run([sys.executable, "-c", FIRST_CALL_CHILD.format(root=ROOT, tp=tp, rp=rp)], capture_output=True,
                             text=True, timeout=300)
        if run.returncode != 0:
            return {"error": run.stderr[-300:]}
        got = json.loads(next(ln for ln in run.stdout.splitlines() if ln.startswith("FIRST "))[6:])
    finally:
        shutil.rmtree(folder, ignore_errors=True)
    first = got["import_ms"] + got["first_main_ms"]
    return {"first_call_ms": round(first, 1), "import_ms": round(got["import_ms"], 1),
            "first_main_ms": round(got["first_main_ms"], 1), "second_main_ms": round(got["second_main_ms"], 1),
            "operator_build_ms": round(max(0.0, got["third_main_new_config_ms"] - got["second_main_ms"]), 1),
            "note": "fresh child process, from `import matchering_amd` to the first stages.main result on the 8-minute pair "
                    "(pageable numpy in, pinned numpy out); second_main_ms = the same call again; operator_build_ms = a call "
                    "with another Config minus that (only the Config's raw -> smooth operator is new)"}


def file_to_file(mg, pair):
    """mg.process (core.py:32-121) on the workload's pair as PCM_16 WAVE files: load, checks, the GPU path,
    one PCM_16 result file.  Integer samples cross PCIe as they are and are decoded / quantised on the GPU
    (mgx_pcm_decode, mgx_peak_count, mgx_pcm_encode).  Never the headline value."""
    import shutil
    import tempfile

    from matchering_amd import audio_io

    target, reference = pair
    folder = tempfile.mkdtemp(prefix="mgx_bench_")
    try:
        tp, rp, op = (os.path.join(folder, n) for n in ("target.wav", "reference.wav", "result.wav"))
        audio_io.write_wav(tp, 0.7 * target, 44100, "PCM_16")
        audio_io.write_wav(rp, 0.9 * reference, 44100, "PCM_16")
        took = []
        for _ in range(3):
            t0 = time.perf_counter()
            mg.process(tp, rp, [mg.pcm16(op)])
            took.append(time.perf_counter() - t0)
    finally:
        shutil.rmtree(folder, ignore_errors=True)
    return {"value": round(target.shape[0] / min(took) / 1e6, 2), "unit": "Msamples/s", "ms": round(min(took) * 1e3, 2),
            "note": "mg.process on the pair as PCM_16 WAVE files -> one PCM_16 file, best of 3 calls (page cache warm)"}


def host_to_host_batch(mg, make_pair, pairs=8, seconds=240.0, lanes=2):
    """Config #4's per-GPU share from host arrays to host arrays: eight four-minute pairs through
    batch.master_many (two lanes = two device handles, one pair's PCIe copies under the other's kernels).
    The inputs sit in pinned host memory, where a loader thread would decode them; PCIe moves 24 B per
    frame (target + reference up, result down), so ~2.3 G frames/s is the bound at 55 GB/s."""
    from matchering_amd import batch
    from matchering_amd.device import pinned

    staged = []
    for k in range(pairs):
        t, r = make_pair(seconds, 44100, pair=200 + k)
        pt, pr = pinned.empty(t.shape), pinned.empty(r.shape)
        pt[...] = t
        pr[...] = r
        staged.append((pt, pr))
    cfg = mg.Config()
    peaks = {}

    def consume(index, triple):            # what a writer thread would do, minus the file: look at it, let it go
        peaks[index] = float(np.abs(triple[0][::4096]).max())

    frames = sum(p[0].shape[0] for p in staged)
    wall = None
    for attempt in range(3):               # the first passes fill the block pools (pinned host, HBM)
        peaks.clear()
        t0 = time.perf_counter()
        batch.master_many(staged, cfg, need_default=True, lanes=lanes, on_result=consume)
        took = time.perf_counter() - t0
        wall = took if wall is None else min(wall, took)
    assert len(peaks) == pairs and all(0.0 < v < 1.0 for v in peaks.values())
    return {"value": round(frames / wall / 1e6, 2), "unit": "Msamples/s", "ms_per_batch": round(wall * 1e3, 2),
            "note": f"{pairs} x {seconds:.0f} s pairs, pinned numpy in -> pinned numpy out, {lanes} lanes"}


def _cpu_worker(seconds, sample_rate, fft, need, pair, use_reference):
    """One process of the all-host-cores figure: its own synthetic pair through the reference's stages.main (when
    staged) or the oracle; returns (seconds of the call, frames)."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    from matchering_amd.synth import make_pair

    try:                        # P processes x a BLAS pool of every core each would only fight: one thread per process
        from threadpoolctl import threadpool_limits

        threadpool_limits(1)
    except Exception:           # noqa: BLE001
        pass
    target, reference = make_pair(seconds, sample_rate, pair=pair)
    if use_reference:
        import warnings

        import build_ref

        mg_ref = build_ref.load()
        from matchering import stages as ref_stages

        cfg = mg_ref.Config(internal_sample_rate=sample_rate, fft_size=fft)
        t64, r64 = target.astype(np.float64), reference.astype(np.float64)
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            t0 = time.perf_counter()
            ref_stages.main(t64, r64, cfg, need_default=need[0], need_no_limiter=need[1], need_no_limiter_normalized=need[2])
            return time.perf_counter() - t0, target.shape[0]
    import mastering_oracle as mo

    cfg = mo.params(internal_sample_rate=sample_rate, fft_size=fft)
    t0 = time.perf_counter()
    mo.master(target, reference, cfg, *need)
    return time.perf_counter() - t0, target.shape[0]


def reference_main(target, reference, sample_rate, fft, need, runs_budget_s=12.0):
    """The reference's OWN stages.main (stages.py:210-272), unmodified, timed on this host: the package as
    oracle/build_ref.py byte-compiled it from /root/reference (oracle/_ref/, shipped with the snapshot -- the tree
    itself does not exist on the GPU box), soundfile / resampy stubbed (file I/O, never reached from stages.main) and
    statsmodels' LOWESS served by the pinned restatement (option B).  Inputs: the GPU path's
    float32 values as float64, the reference's native type.  Returns (seconds of the best run, runs, output, info) or
    None when no staged reference is here."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    try:
        import build_ref

        if not build_ref.reference_present():
            # a Python reference does not travel to the GPU box in any form: whatever byte code a snapshot may have
            # carried along is not used there -- the reference is timed where its tree is (the build container)
            return None, "no /root/reference on this box"
        mg_ref = build_ref.load()
    except Exception as exc:                                 # noqa: BLE001 -- not staged / wrong interpreter: the port stands in
        return None, repr(exc)[:200]
    import warnings

    from matchering import stages as ref_stages

    cfg = mg_ref.Config(internal_sample_rate=sample_rate, fft_size=fft)
    t64, r64 = target.astype(np.float64), reference.astype(np.float64)
    runs, out = [], None
    t_all = time.perf_counter()
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        while len(runs) < 2 or (time.perf_counter() - t_all < runs_budget_s and len(runs) < 4):
            t0 = time.perf_counter()
            out = ref_stages.main(t64.copy(), r64.copy(), cfg, need_default=need[0], need_no_limiter=need[1],
                                  need_no_limiter_normalized=need[2])
            runs.append(time.perf_counter() - t0)
    m = build_ref.manifest()
    build_ref.unload()                   # (the stub modules and the staged package leave this process again)
    import platform

    import scipy

    info = {"package": f"sergree/matchering {m.get('version')}", "modules": len(m.get("files", {})),
            "python": platform.python_version(), "numpy": np.__version__, "scipy": scipy.__version__,
            "lowess": "oracle/mastering_oracle.py lowess_it0 (statsmodels is not installed; pinned to the compiled "
                      "statsmodels 0.12.2 to 1e-12 by tests/golden)"}
    return (min(runs), runs, next(o for o in out if o is not None), info), None


def cpu_model():
    try:
        with open("/proc/cpuinfo") as fh:
            for ln in fh:
                if ln.lower().startswith("model name"):
                    return ln.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def cpu_baseline(wl, name, all_cores=True):
    """The reference's CPU path on this box's host cores, beside every GPU number:
    (i) ``kind: "reference"`` -- sergree/matchering's own ``stages.main`` (reference_main above) on the workload's first
    pair, one process, what a matchering user gets; the oracle (``port``: oracle/mastering_oracle.py, the numpy/scipy
    restatement the parity tests use) is timed on the same pair and kept beside it; when no staged reference is
    here the port IS the baseline and says so (``kind: "port"``);  (ii) the all-host-cores figure: P concurrent
    processes of one 60-second pair each (the port: the processes are the oracle's)."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import mastering_oracle as mo

    target, reference = wl.host_pair
    ocfg = mo.params(internal_sample_rate=wl.sample_rate, fft_size=wl.fft)
    need = (True, False, False) if wl.want_limiter else (False, True, False)
    runs = []
    result = None
    t_all = time.perf_counter()
    while len(runs) < 2 or (time.perf_counter() - t_all < 8.0 and len(runs) < 4):
        t0 = time.perf_counter()
        result = mo.master(target, reference, ocfg, *need)
        runs.append(time.perf_counter() - t0)
    result = next(r for r in result if r is not None)
    cpu_s = min(runs)
    n = target.shape[0]
    port = {"value": round(n / cpu_s / 1e6, 3), "unit": "Msamples/s", "cores": 1, "seconds": round(cpu_s, 2),
            "runs": len(runs), "what": "oracle/mastering_oracle.py (numpy/scipy float64 restatement of stages.main), one thread"}
    host = {"cpu": cpu_model(), "logical_cores": os.cpu_count()}
    ref, why_not = reference_main(target, reference, wl.sample_rate, wl.fft, need)
    if ref is not None:
        ref_s, ref_runs, ref_out, info = ref
        gap = ref_out - result
        out = {"value": round(n / ref_s / 1e6, 3), "unit": "Msamples/s", "cores": 1, "kind": "reference",
               "seconds": round(ref_s, 2), "runs": len(ref_runs), "host": host, "reference": info,
               "sample": f"the workload's first pair ({n} frames, float32 values as float64), best of {len(ref_runs)} runs of "
                         f"matchering.stages.main (stages.py:210-272, unmodified; {sum(ref_runs):.0f} s of CPU work), one "
                         f"process: the reference is single-threaded",
               "port": dict(port, port_over_reference_seconds=round(cpu_s / ref_s, 3),
                            max_abs_difference_from_reference=float(np.abs(gap).max())),
               "note": "kind 'reference': sergree/matchering's own stages.main, byte-compiled from /root/reference by "
                       "oracle/build_ref.py under oracle/_ref/ (git- and gpurun-ignored: only where /root/reference exists); "
                       "`port` is the oracle on the same pair"}
    else:
        out = dict(port, kind="port", host=host,
                   sample=f"the workload's first pair ({n} frames), best of {len(runs)} runs of oracle/mastering_oracle.py "
                          f"({sum(runs):.0f} s of CPU work)",
                   note="kind 'port': the oracle (oracle/mastering_oracle.py), not sergree/matchering itself -- a Python "
                        "reference does not travel to the GPU box in any form; port_vs_reference = both timed on ONE host "
                        "(the build container, tools/cpu_port_vs_reference.py) and the largest difference of their outputs "
                        f"on the 8-minute pair ({why_not})")
        ratio_path = os.path.join(ROOT, "profiles", "cpu_port_vs_reference.json")
        if os.path.exists(ratio_path):
            with open(ratio_path) as fh:
                out["port_vs_reference"] = json.load(fh)
    if not all_cores:
        return out, result
    try:
        import concurrent.futures as cf

        procs = max(1, min(os.cpu_count() or 1, 32))
        sample_seconds = 60.0
        with cf.ProcessPoolExecutor(max_workers=procs) as pool:
            done = list(pool.map(_cpu_worker, [sample_seconds] * procs, [wl.sample_rate] * procs, [wl.fft] * procs,
                                 [need] * procs, range(procs), [ref is not None] * procs))
        slowest = max(d[0] for d in done)
        out["all_cores"] = {"value": round(sum(d[1] for d in done) / slowest / 1e6, 3), "unit": "Msamples/s",
                            "processes": procs, "kind": "reference" if ref is not None else "port",
                            "sample": f"{procs} concurrent processes (of {os.cpu_count()} logical cores), one "
                                      f"{sample_seconds:.0f} s pair each, frames of all / time of the slowest "
                                      f"({slowest:.1f} s; each process times its own call of stages.main)"}
    except Exception as exc:           # noqa: BLE001
        out["all_cores"] = {"error": str(exc)[:200]}
    return out, result


if __name__ == "__main__":
    main()
