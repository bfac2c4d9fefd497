"""Batch front end: many independent target/reference pairs on one or several MI355X.

The reference has no batch API (``core.process`` masters exactly one pair, core.py:32-121; batches
are left to the external matchering-cli, README.md:146-147).  Pairs share nothing, so the batch
axis shards without any data-path collective:

* across GPUs: pair ``i`` belongs to rank ``i mod world_size``, one process per GPU.  Ranks are read
  from the environment ``torch.distributed.run`` (or any launcher) sets -- RANK, WORLD_SIZE,
  LOCAL_RANK -- and no rank ever talks to another one here;
* inside a rank: ``lanes`` device handles (each its own HIP stream and workspace) are fed by one
  thread each.  A pair's kernels form a dependent chain with short single-workgroup links (FIR
  design, level-correction decisions, the limiter's look-back waits) and its PCIe copies run in one
  direction at a time; the other lanes fill those holes with other pairs' work (three by default: eight
  resident 4-minute pairs take 2.6 / 2.1 / 1.85 ms with one / two / three; ``MAX_LANES`` says why not many);
* around the GPU: reading and writing audio files is host work, so ``process_batch`` keeps ``io_threads``
  loaders ahead of the lanes and writes results behind them.  Integer PCM goes to the GPU as the file
  holds it and comes back quantised (``stages.main``), which leaves the host little more than file I/O.

``master_many`` works on arrays in memory, ``process_batch`` on files with the reference's
``Result`` objects.  Command line (one rank)::

    python -m matchering_amd.batch jobs.json

and on a node with 8 GPUs::

    python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 \\
        -m matchering_amd.batch jobs.json

``jobs.json`` is a list of ``{"target": path, "reference": path, "results": [{"file": path,
"subtype": "PCM_16", "use_limiter": true, "normalize": true}, ...]}``.
"""

import json
import os
import queue
import sys
import threading

import numpy as np
from concurrent.futures import ThreadPoolExecutor

from .config import Config
from .log import Code, ModuleError
from .results import Result


def rank_and_world(rank=None, world_size=None):
    """(rank, world_size, local_rank) from the arguments or the launcher's environment."""
    r = int(os.environ.get("RANK", "0")) if rank is None else int(rank)
    w = int(os.environ.get("WORLD_SIZE", "1")) if world_size is None else int(world_size)
    local = int(os.environ.get("LOCAL_RANK", str(r)))
    if not 0 <= r < w:
        raise ValueError(f"rank {r} outside world of {w}")
    return r, w, local


def shard(items, rank, world_size):
    """Indices of the items rank ``rank`` owns: ``i mod world_size == rank``."""
    return list(range(rank, len(items), world_size))


class _Lanes:
    """``lanes`` worker threads, each bound to its own device handle, draining one job queue."""

    def __init__(self, make_worker, lanes):
        # bounded: ``submit`` blocks while every lane is busy and one job per lane is waiting, so whoever
        # feeds the lanes (the file loaders of process_batch) runs only that far ahead of the GPU
        self.jobs = queue.Queue(maxsize=max(1, lanes))
        self.failure = None
        self.threads = [threading.Thread(target=self._run, args=(make_worker, lane), daemon=True)
                        for lane in range(lanes)]
        for t in self.threads:
            t.start()

    def _run(self, make_worker, lane):
        worker = None
        while True:
            item = self.jobs.get()
            if item is None:
                return
            index, payload, done = item
            try:
                if self.failure is not None:
                    raise RuntimeError("batch aborted")
                if worker is None:
                    worker = make_worker(lane)
                done(index, worker(payload), None)
            except BaseException as exc:           # recorded and re-raised by close()
                if self.failure is None:
                    self.failure = exc
                done(index, None, exc)

    def submit(self, index, payload, done):
        self.jobs.put((index, payload, done))

    def close(self):
        for _ in self.threads:
            self.jobs.put(None)
        for t in self.threads:
            t.join()
        if self.failure is not None:
            raise self.failure


_lane_devices = {}
_lane_devices_lock = threading.Lock()


# More lanes than this are not used: k_correction_tail keeps its <= 128 workgroups resident together (48 KB of
# LDS each: 768 fit the chip), and the tails of seven pairs dispatched side by side could each hold part of
# the slots and wait for the rest.
MAX_LANES = 6


_lane_choice = {}


def lanes_allowed(world_size=1):
    """MAX_LANES is a budget per GPU, not per process: ranks that share one GPU split it."""
    from .device import device_count

    sharing = max(1, -(-int(world_size) // max(1, device_count())))
    if sharing > 1:
        ranks_share_a_gpu()
    return max(1, MAX_LANES // sharing)


def ranks_share_a_gpu():
    """More rank PROCESSES than GPUs: their limiter launches can be resident on one chip together, where chunks dealt by
    workgroup number may wait for each other for ever (mgx.hip, LimiterChain -- handles of one process are chained,
    processes cannot see each other).  The library then deals chunks by an atomic ticket: slower by a fifth, safe with
    any neighbour.  Read by the library at every limiter launch."""
    os.environ.setdefault("MGX_LIMIT_TICKETS", "1")


_measuring = threading.Lock()


def choose_lanes(device_index=0, candidates=(2, 3), seconds=120.0, pairs=6, master=None, world_size=1):
    """How many device handles a batch should run on THIS GPU: measured, once per process and GPU.

    Boxes of the pool disagree: on some, three handles beat two by 10 % (1.87 vs 2.11 ms for eight resident
    4-minute pairs), on others three LOSE to two by 13 % (2.93 vs 2.60 ms; profiles/r02_lanes.txt,
    profiles/r02_g_bench_slow_box.json, BENCH_r02.json) -- so the number is not a constant of the code.  A
    short synthetic batch (``pairs`` resident pairs of ``seconds`` each -- six: a multiple of both candidate
    counts, so neither is handed an uneven share --, full pipeline) is timed through
    each candidate count and the fastest wins; the decision and both timings are kept in
    ``lane_choice_report(device_index)``.  Costs ~1 s of host time for the synthetic material and a few
    milliseconds of GPU time.  ``MGX_LANES=n`` skips the measurement; candidates above the per-GPU budget of
    ``lanes_allowed(world_size)`` are not tried (ranks that share a GPU would otherwise measure with more handles
    than they may use); one measurement at a time per process, each lane's device locked while it runs, and the
    blocks it recycled are given back afterwards."""
    if master is not None:                       # a stand-in for the GPU (CPU tests): nothing to measure
        return min(candidates[-1], MAX_LANES)
    forced = os.environ.get("MGX_LANES", "")
    if forced.isdigit() and int(forced) > 0:
        return min(int(forced), lanes_allowed(world_size))
    budget = lanes_allowed(world_size)
    candidates = tuple(c for c in candidates if c <= budget) or (budget,)
    if len(candidates) == 1:
        return candidates[0]
    with _measuring:
        with _lane_devices_lock:
            if device_index in _lane_choice:
                return min(_lane_choice[device_index]["lanes"], budget)
        import time
        from contextlib import ExitStack

        from .synth import make_pair

        native = Config().to_native()
        host = [make_pair(seconds, 44100, pair=300 + k) for k in range(pairs)]
        timings = {}
        for count in candidates:
            devs = [lane_device(device_index, lane) for lane in range(count)]
            with ExitStack() as held:
                for d in devs:
                    held.enter_context(d.lock)
                jobs = []
                for k, (t, r) in enumerate(host):
                    d = devs[k % count]
                    jobs.append((d, d.upload(t), t.shape[0], d.upload(r), r.shape[0], d.alloc(t.shape[0] * 8)))

                def step():
                    for d, t, n, r, nr, out in jobs:
                        d.master(t, n, r, nr, native, result=out, want_report=False)

                def sync():
                    for d in devs:
                        d.synchronize()

                def settled():
                    # the one-time switch of a handle to its safe mode (a GPU shared with somebody else's kernels,
                    # MGX_ERR_RETRY) belongs in front of the measurement, once per handle and kind of wait
                    from ._native import MgxError

                    for _ in range(2 * len(devs) + 1):
                        step()
                        try:
                            sync()
                            return
                        except MgxError as exc:
                            if not exc.retry:
                                raise
                    raise RuntimeError("the lanes' handles keep losing their launches")

                settled()
                best = None
                for _ in range(3):
                    t0 = time.perf_counter()
                    step()
                    step()
                    try:
                        sync()
                    except Exception as exc:             # (a switch in mid-measurement: this sample does not count)
                        if not getattr(exc, "retry", False):
                            raise
                        settled()
                        continue
                    took = (time.perf_counter() - t0) / 2
                    best = took if best is None else min(best, took)
                if best is None:
                    best = float("inf")
                timings[count] = best
                for d, t, n, r, nr, out in jobs:
                    for b in (t, r, out):
                        b.release()
                for d in devs:
                    d.trim()
        chosen = min(timings, key=timings.get)
        with _lane_devices_lock:
            _lane_choice[device_index] = {"lanes": chosen, "ms_per_batch": {str(k): round(v * 1e3, 3) for k, v in timings.items()},
                                          "batch": f"{pairs} resident pairs of {seconds:.0f} s"}
        return chosen


def lane_choice_report(device_index=0):
    return dict(_lane_choice.get(device_index, {}))


def lane_device(device_index, lane):
    """The device handle of lane ``lane`` on GPU ``device_index``: created once per process and kept --
    with it its FIR plans, workspaces and recycled HBM blocks, which a batch should not pay for twice."""
    from .device import Device

    with _lane_devices_lock:
        key = (device_index, lane)
        if key not in _lane_devices:
            _lane_devices[key] = Device(device_index)
        return _lane_devices[key]


def _device_worker(device_index, lane, config, needs, master, encodings=None):
    """A callable mastering one (target, reference) array pair on the lane's own device handle.
    ``encodings``: stages.main's (integer PCM renderings straight from the GPU)."""
    if master is not None:                         # tests inject a stand-in for the GPU
        if encodings is None or not any(encodings):
            return lambda pair: master(pair[0], pair[1], config, *needs)
        return lambda pair: master(pair[0], pair[1], config, *needs, encodings=encodings)
    from .stages import main

    dev = lane_device(device_index, lane)
    return lambda pair: main(pair[0], pair[1], config, *needs, device=dev, encodings=encodings)


def master_many(pairs, config=None, need_default=True, need_no_limiter=False,
                need_no_limiter_normalized=False, device_index=0, lanes=None, master=None, on_result=None):
    """``stages.main`` over a list of (target, reference) arrays on ONE GPU, ``lanes`` pairs in flight
    (``None``: as many as ``choose_lanes`` measures to be best on this GPU).

    Returns the list of result triples in the order of ``pairs``.  Results are bit-identical to
    calling ``stages.main`` pair by pair: lanes only change when work is submitted, never what is
    computed.  With ``on_result(index, triple)`` every triple is handed over as it completes (from a lane
    thread) and NOT kept: the pinned host blocks the results live in are recycled as soon as the
    consumer lets go of them, and the returned list holds ``None``."""
    config = config if config is not None else Config()
    needs = (need_default, need_no_limiter, need_no_limiter_normalized)
    out = [None] * len(pairs)

    def done(index, value, exc):
        if on_result is not None and exc is None:
            on_result(index, value)
        else:
            out[index] = value

    if lanes is None:
        lanes = choose_lanes(device_index, master=master) if len(pairs) > 2 else 2
    pool = _Lanes(lambda lane: _device_worker(device_index, lane, config, needs, master),
                  max(1, min(lanes, MAX_LANES, len(pairs) or 1)))
    for i, pair in enumerate(pairs):
        pool.submit(i, pair, done)
    pool.close()
    return out


def master_album(targets, reference, config=None, rank=None, world_size=None, device_index=None, root=0,
                 exchange=None, need_default=True, need_no_limiter=False, need_no_limiter_normalized=False):
    """Album mode across the ranks of one node: ONE matching FIR for every track.

    Rank ``root`` masters ``targets[0]`` against ``reference`` the ordinary way; the FIR pair it designed
    (2 x fft_size float32, 32 KiB at the default size) is the only thing that crosses xGMI -- one
    ``ncclBroadcast`` over RCCL -- and every rank then masters its share of the targets (track i -> rank
    i mod world) with that FIR given (``mgx_master_with_fir``), levels still matched per track.  Returns
    ``{index: triple}`` for this rank's tracks.  ``exchange(payload, size)`` must hand rank 0's 128-byte
    RCCL id to all ranks (bench.Ranks.broadcast_bytes); not needed with one rank."""
    from .stages import main

    config = config if config is not None else Config()
    r, w, local = rank_and_world(rank, world_size)
    dev = lane_device(local if device_index is None else device_index, 0)
    needs = (need_default, need_no_limiter, need_no_limiter_normalized)
    count = 2 * config.fft_size
    mine = shard(targets, r, w)
    results = {}
    with dev.lock:
        fir = dev.alloc(count * 4)
        if r == root:                                           # the design pass; its result is track 0's
            first = main(targets[0], reference, config, *needs, device=dev)
            taps_ptr, taps = dev.last_fir()
            assert taps == config.fft_size
            designed = dev.download(taps_ptr, (count,))
            fir.release()
            fir = dev.upload(designed)
            if 0 in mine:
                results[0] = first
        dev.comm_init(r, w, exchange)
        try:
            dev.comm_broadcast(fir, count, root)
            dev.synchronize()
        finally:
            dev.comm_destroy()
        for i in mine:
            if i not in results:
                results[i] = main(targets[i], reference, config, *needs, device=dev, fir=fir)
        fir.release()
    return results


def _peaks_on_the_lane(target, device_index, lane, config, master):
    """The part of checker.check that ``_load_job`` left out for an integer PCM target: upload it on the
    lane's device, take count_max_peaks there (mgx_peak_count), warn as the reference would; the resident
    frames go on to stages.main.  With a stand-in for the GPU the statistics are taken on the host."""
    from .audio_io import unpack24
    from .checker import count_max_peaks, peak_warnings

    if master is not None:
        peak_warnings(count_max_peaks(unpack24(target) if target.dtype == np.uint8 else target), config)
        return target
    from .device import DeviceFrames

    dev = lane_device(device_index, lane)
    with dev.lock:
        frames = DeviceFrames(dev.upload_frames(target), target.shape[0])
        peaks = dev.peak_count(frames, 2 * target.shape[0])
    peak_warnings(peaks, config)
    return frames


def _wanted_encodings(results):
    from .core import _wanted_encodings as of_results

    return of_results(results)


def _needs_of(results):
    return (any(r.use_limiter for r in results),
            any(not r.use_limiter and not r.normalize for r in results),
            any(not r.use_limiter and r.normalize for r in results))


def _load_job(job, config):
    """Load + check both files of a job (core.py:52-74), on a host thread.  Returns (target, reference,
    deferred): with ``deferred`` the target is integer PCM that goes to the GPU as it is, and its peak
    statistics (checker.py:118-130) are left to the lane that masters it (``mgx_peak_count``)."""
    from .audio_io import load, pcm_channels
    from .checker import LATER, check, check_equality
    from .utils import get_temp_folder

    temp_folder = config.temp_folder if config.temp_folder else get_temp_folder(job["results"])
    # (pcm=True: 16/24/32-bit WAVE samples stay integers up to the GPU, as in core.process)
    target, rate_t = load(job["target"], "target", temp_folder, pcm=True)
    deferred = ((target.dtype.kind in "iu" or target.dtype == np.float32) and pcm_channels(target) == 2
                and rate_t == config.internal_sample_rate)
    target, rate_t = check(target, rate_t, config, "target", peaks=LATER if deferred else None)
    reference, rate_r = load(job["reference"], "reference", temp_folder, pcm=True)
    reference, rate_r = check(reference, rate_r, config, "reference")
    if not config.allow_equality:
        check_equality(target, reference)
    if (not (rate_t == rate_r == config.internal_sample_rate)
            or not (pcm_channels(target) == pcm_channels(reference) == 2)
            or not (target.shape[0] > config.fft_size and reference.shape[0] > config.fft_size)):
        raise ModuleError(Code.ERROR_VALIDATION)
    return target, reference, deferred


def _save_job(job, triple, config):
    from .audio_io import save

    result, plain, normalized = triple
    for wanted in job["results"]:
        chosen = result if wanted.use_limiter else (normalized if wanted.normalize else plain)
        save(wanted.file, chosen, config.internal_sample_rate, wanted.subtype)


def process_batch(jobs, config=None, rank=None, world_size=None, device_index=None, lanes=None, io_threads=4,
                  master=None):
    """``process`` for a list of jobs, this rank's share only.

    ``jobs``: dicts with "target", "reference" (paths) and "results" (list of ``Result``).  Returns
    the indices of the jobs this rank mastered.  The first failing job aborts the rank's batch and
    its exception is re-raised (after the jobs already in flight have finished)."""
    config = config if config is not None else Config()
    r, w, local = rank_and_world(rank, world_size)
    mine = shard(jobs, r, w)
    for i in mine:
        if not jobs[i].get("results"):
            raise RuntimeError("The result list is empty")
    device_index = local if device_index is None else device_index
    savers = []

    with ThreadPoolExecutor(max_workers=max(1, io_threads)) as io:
        def worker_for(lane):
            workers = {}

            def run(item):
                index, (target, reference, deferred) = item
                needs = _needs_of(jobs[index]["results"])
                key = (needs, _wanted_encodings(jobs[index]["results"]))
                if key not in workers:
                    workers[key] = _device_worker(device_index, lane, config, needs, master, key[1])
                if deferred:
                    target = _peaks_on_the_lane(target, device_index, lane, config, master)
                return workers[key]((target, reference))
            return run

        if lanes is None:
            lanes = choose_lanes(device_index, master=master, world_size=w) if len(mine) > 2 else 2
        lanes = min(max(1, lanes), MAX_LANES if master is not None else lanes_allowed(w))
        pool = _Lanes(worker_for, lanes)
        # Host memory stays bounded whatever the batch size: at most `io_threads` decoded pairs wait for
        # the lanes (the loaders are started one by one as their predecessors are consumed, and the lane
        # queue is bounded), and at most `lanes + io_threads` mastered triples wait for their writers.
        unsaved = threading.Semaphore(max(1, lanes) + max(1, io_threads))

        def done(index, triple, exc):
            if exc is None:
                unsaved.acquire()
                future = io.submit(_save_job, jobs[index], triple, config)
                future.add_done_callback(lambda _f: unsaved.release())
                savers.append(future)

        ahead = {k: io.submit(_load_job, jobs[i], config) for k, i in enumerate(mine[:io_threads])}
        nxt = len(ahead)
        try:
            for k, i in enumerate(mine):
                arrays = ahead.pop(k).result()            # (popped: the decoded arrays live on only in the job)
                if nxt < len(mine):
                    ahead[nxt] = io.submit(_load_job, jobs[mine[nxt]], config)
                    nxt += 1
                pool.submit(i, (i, arrays), done)
                del arrays
        finally:
            pool.close()
        for s in savers:
            s.result()
    if master is None:                       # a batch of varied track lengths leaves many size classes behind
        from .device import pinned

        for lane in range(lanes):
            lane_device(device_index, lane).trim()
        pinned.trim()
    return mine


def jobs_from_json(path):
    with open(path) as fh:
        raw = json.load(fh)
    jobs = []
    for item in raw:
        results = [Result(r["file"], subtype=r.get("subtype", "PCM_16"), use_limiter=r.get("use_limiter", True),
                          normalize=r.get("normalize", True)) for r in item["results"]]
        jobs.append({"target": item["target"], "reference": item["reference"], "results": results})
    return jobs


def main(argv=None):
    argv = sys.argv[1:] if argv is None else argv
    if len(argv) != 1:
        raise SystemExit(__doc__)
    jobs = jobs_from_json(argv[0])
    r, w, _ = rank_and_world()
    done = process_batch(jobs)
    print(f"rank {r}/{w}: mastered {len(done)} of {len(jobs)} pairs")


if __name__ == "__main__":
    main()
