"""Batch front end (matchering_amd/batch.py): sharding by rank,
lanes, file pipeline.  On the CPU the GPU is replaced by the float64 oracle through the ``master``
hook (the hook exists for exactly this: the product path has no CPU implementation)."""

import json
import os
import threading

import numpy as np
import pytest

import mastering_oracle as mo
import matchering_amd as mg
from matchering_amd import audio_io, batch
from matchering_amd.synth import synth


def _oracle_master(target, reference, config, need_default, need_no_limiter, need_no_limiter_normalized,
                   encodings=None):
    """stages.main's contract on the CPU oracle: integer PCM frames in are decoded, renderings asked for in
    an integer subtype come back quantised (audio_io's codec = what mgx_pcm_decode / _encode do)."""
    ocfg = mo.params(internal_sample_rate=config.internal_sample_rate, fft_size=config.fft_size)
    out = mo.master(audio_io.pcm_to_float(np.asarray(target), np.float64),
                    audio_io.pcm_to_float(np.asarray(reference), np.float64), ocfg,
                    need_default, need_no_limiter, need_no_limiter_normalized)
    coded = []
    for o, fmt in zip(out, encodings or (None, None, None)):
        if o is None or fmt is None:
            coded.append(None if o is None else o.astype(np.float32))
        else:
            bits = int(fmt[4:])
            q = audio_io._quantise(o.astype(np.float32), bits)
            coded.append(audio_io._pack24(q).reshape(-1, 6) if bits == 24 else q.reshape(o.shape))
    return tuple(coded)


def _pairs(count, seconds=1.5, rate=8000):
    out = []
    for b in range(count):
        t = (0.5 * synth(seconds, rate, 1 + 2 * b)).astype(np.float32)
        r = np.clip(2.5 * synth(seconds, rate, 2 + 2 * b), -1, 1).astype(np.float32)
        out.append((t, r))
    return out


def test_shard_is_a_partition():
    items = list(range(13))
    for world in (1, 2, 3, 8, 16):
        seen = sorted(i for r in range(world) for i in batch.shard(items, r, world))
        assert seen == items
        assert all(i % world == r for r in range(world) for i in batch.shard(items, r, world))


def test_rank_from_environment(monkeypatch):
    monkeypatch.setenv("RANK", "3")
    monkeypatch.setenv("WORLD_SIZE", "8")
    monkeypatch.setenv("LOCAL_RANK", "3")
    assert batch.rank_and_world() == (3, 8, 3)
    assert batch.rank_and_world(1, 2)[:2] == (1, 2)
    with pytest.raises(ValueError):
        batch.rank_and_world(2, 2)


def test_master_many_keeps_order_and_matches_single_runs():
    cfg = mg.Config(internal_sample_rate=8000, fft_size=256, max_piece_size=2)
    pairs = _pairs(5)
    seen_threads = set()

    def master(t, r, c, *needs):
        seen_threads.add(threading.get_ident())
        return _oracle_master(t, r, c, *needs)

    many = batch.master_many(pairs, cfg, need_default=True, need_no_limiter=True, lanes=2, master=master)
    assert len(many) == len(pairs) and 1 <= len(seen_threads) <= 2
    for (t, r), triple in zip(pairs, many):
        want = _oracle_master(t, r, cfg, True, True, False)
        assert np.array_equal(triple[0], want[0]) and np.array_equal(triple[1], want[1]) and triple[2] is None


def test_master_many_reraises_the_first_failure():
    cfg = mg.Config(internal_sample_rate=8000, fft_size=256, max_piece_size=2)

    def master(t, r, c, *needs):
        raise RuntimeError("device lost")

    with pytest.raises(RuntimeError, match="device lost"):
        batch.master_many(_pairs(3), cfg, master=master)


def test_process_batch_files_two_ranks(tmp_path):
    rate = 8000
    cfg = mg.Config(internal_sample_rate=rate, fft_size=256, max_piece_size=2)
    jobs = []
    for b, (t, r) in enumerate(_pairs(4, seconds=2.0, rate=rate)):
        tp, rp = str(tmp_path / f"t{b}.wav"), str(tmp_path / f"r{b}.wav")
        kind = {1: "PCM_16", 2: "PCM_24"}.get(b, "FLOAT")     # jobs 1, 2: integer files stay integers up to the device
        audio_io.write_wav(tp, t, rate, kind)
        audio_io.write_wav(rp, r, rate, kind)
        jobs.append({"target": tp, "reference": rp,
                     "results": [mg.Result(str(tmp_path / f"out{b}.wav"), kind),
                                 mg.Result(str(tmp_path / f"plain{b}.wav"), "FLOAT", use_limiter=False, normalize=False)]})
    done = []
    for rank in (0, 1):
        done += batch.process_batch(jobs, cfg, rank=rank, world_size=2, lanes=2, io_threads=2, master=_oracle_master)
    assert sorted(done) == [0, 1, 2, 3]
    for b, job in enumerate(jobs):
        t, _ = audio_io.read_wav(job["target"])
        r, _ = audio_io.read_wav(job["reference"])
        want = _oracle_master(t, r, cfg, True, True, False)
        got, _ = audio_io.read_wav(str(tmp_path / f"out{b}.wav"))
        plain, _ = audio_io.read_wav(str(tmp_path / f"plain{b}.wav"))
        step = {1: 1.6 / 32768, 2: 1.6 / (1 << 23) + 1e-6}.get(b, 1e-6)
        assert np.abs(got - want[0]).max() <= step and np.abs(plain - want[1]).max() <= 1e-6


def test_jobs_from_json(tmp_path):
    path = tmp_path / "jobs.json"
    path.write_text(json.dumps([{"target": "a.wav", "reference": "b.wav",
                                 "results": [{"file": "o.wav", "subtype": "PCM_24", "use_limiter": False}]}]))
    jobs = batch.jobs_from_json(str(path))
    assert jobs[0]["target"] == "a.wav" and jobs[0]["results"][0].subtype == "PCM_24"
    assert jobs[0]["results"][0].use_limiter is False and jobs[0]["results"][0].normalize is True


@pytest.mark.gpu
def test_gpu_lanes_are_bit_identical_to_single_runs():
    from matchering_amd import stages

    cfg = mg.Config()
    pairs = []
    for b in range(4):
        t = (0.5 * synth(6.0, 44100, 1 + 2 * b)).astype(np.float32)
        r = np.clip(2.5 * synth(5.0, 44100, 2 + 2 * b), -1, 1).astype(np.float32)
        pairs.append((t, r))
    many = batch.master_many(pairs, cfg, need_default=True, need_no_limiter=True, lanes=2)
    for (t, r), triple in zip(pairs, many):
        want = stages.main(t, r, cfg, need_default=True, need_no_limiter=True)
        assert np.array_equal(triple[0], want[0]) and np.array_equal(triple[1], want[1])


@pytest.mark.gpu
def test_gpu_limiters_of_two_handles_never_wait_for_each_other():
    """Two device handles of one process, each with a queue of two-minute pairs whose limiter grids (1477 chunks) are
    larger than the chip holds at once (1024): left to themselves the two limiter launches would be resident together,
    each XCD's slots full of one launch's waiting workgroups while the other launch's lowest chunk cannot start there
    (k_limit deals chunks by workgroup number).  The library chains limiter launches of a device through an event per
    handle (mgx.hip, LimiterChain): every call completes, no look-back wait expires, results equal a lone run's."""
    from matchering_amd.device import Device

    cfg = mg.Config()
    native = cfg.to_native()
    t = np.clip(1.4 * synth(120.0, 44100, 71), -1, 1).astype(np.float32)       # hot: every chunk is busy
    r = np.clip(2.5 * synth(100.0, 44100, 72), -1, 1).astype(np.float32)
    lone = Device(0)
    try:
        td, rd, out = lone.upload(t), lone.upload(r), lone.alloc(t.shape[0] * 8)
        lone.master(td, t.shape[0], rd, r.shape[0], native, result=out, want_report=False)
        want = np.array(lone.download(out, (t.shape[0], 2)))
    finally:
        lone.close()
    a, b = Device(0), Device(0)
    try:
        bufs = []
        for d in (a, b):
            bufs.append((d.upload(t), d.upload(r), d.alloc(t.shape[0] * 8)))
        a.synchronize(), b.synchronize()
        for _ in range(12):                                                      # queued without waiting: the lanes of a batch
            for d, (td, rd, out) in zip((a, b), bufs):
                d.master(td, t.shape[0], rd, r.shape[0], native, result=out, want_report=False)
        for d, (td, rd, out) in zip((a, b), bufs):
            d.synchronize()                                                      # (raises if a bounded wait expired)
            assert np.array_equal(np.array(d.download(out, (t.shape[0], 2))), want)
    finally:
        a.close(), b.close()


@pytest.mark.gpu
def test_gpu_album_mode_one_broadcast_fir_for_every_track():
    """batch.master_album on one rank: the FIR designed on track 0 goes through ncclBroadcast (RCCL, a
    one-rank communicator here) and is applied to every track; levels are matched per track.  Track 0
    equals ordinary mastering bit for bit, the others equal the oracle run with that FIR given."""
    import mastering_oracle as mo
    from matchering_amd import stages

    cfg = mg.Config(max_piece_size=2.0)
    reference = np.clip(2.5 * synth(5.0, 44100, 40), -1, 1).astype(np.float32)
    targets = [(0.5 * synth(6.0 + b, 44100, 41 + b, corner=1500.0 + 700.0 * b)).astype(np.float32) for b in range(3)]
    album = batch.master_album(targets, reference, cfg, rank=0, world_size=1, need_default=True, need_no_limiter=True)
    assert sorted(album) == [0, 1, 2]
    plain = stages.main(targets[0], reference, cfg, need_default=True, need_no_limiter=True)
    assert np.array_equal(album[0][0], plain[0]) and np.array_equal(album[0][1], plain[1])
    tr = {}
    ocfg = mo.params(max_piece_size=2.0)
    mo.master(targets[0], reference, ocfg, True, True, False, trace=tr)
    for b in (1, 2):
        want = mo.master(targets[b], reference, ocfg, True, True, False, fir=(tr["fir_mid"], tr["fir_side"]))
        own = mo.master(targets[b], reference, ocfg, True, True, False)
        for mine, ref, other in zip(album[b][:2], want[:2], own[:2]):
            err = float(np.sqrt(np.mean((mine.astype(np.float64) - ref) ** 2)))
            assert err <= 1e-5
            assert float(np.sqrt(np.mean((ref - other) ** 2))) > 10 * err     # it IS another FIR than the track's own


def _batch_rank(rank, world, tmp, port, queue):
    import os
    import sys

    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK="0")     # both ranks on the one GPU
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import matchering_amd as mg2
    from matchering_amd import batch as b2

    cfg = mg2.Config(max_piece_size=2.0)
    jobs = [{"target": os.path.join(tmp, f"t{k}.wav"), "reference": os.path.join(tmp, f"r{k}.wav"),
             "results": [mg2.Result(os.path.join(tmp, f"out{k}.wav"), "FLOAT")]} for k in range(5)]
    queue.put((rank, b2.process_batch(jobs, cfg, lanes=2, io_threads=2)))


@pytest.mark.gpu
def test_gpu_process_batch_two_ranks_share_the_gpu(tmp_path):
    """The N > 1 path of the batch front end with real kernels: two PROCESSES (rank 0 and 1 of a world of
    2, both on the one visible GPU) take their shares of five jobs -- pair i -> rank i mod 2 -- and every
    result file matches the oracle."""
    import multiprocessing as mp

    import mastering_oracle as mo

    rate = 44100
    pairs = []
    for k in range(5):
        t = (0.5 * synth(4.0 + 0.5 * k, rate, 61 + 2 * k)).astype(np.float32)
        r = np.clip(2.5 * synth(4.0, rate, 62 + 2 * k), -1, 1).astype(np.float32)
        audio_io.write_wav(str(tmp_path / f"t{k}.wav"), t, rate, "FLOAT")
        audio_io.write_wav(str(tmp_path / f"r{k}.wav"), r, rate, "FLOAT")
        pairs.append((t, r))
    ctx = mp.get_context("spawn")
    queue = ctx.Queue()
    procs = [ctx.Process(target=_batch_rank, args=(rank, 2, str(tmp_path), 0, queue)) for rank in range(2)]
    for p in procs:
        p.start()
    shares = dict(queue.get(timeout=600) for _ in procs)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    assert shares == {0: [0, 2, 4], 1: [1, 3]}
    for k, (t, r) in enumerate(pairs):
        got, _ = audio_io.read_wav(str(tmp_path / f"out{k}.wav"))
        want = mo.master(t, r, mo.params(max_piece_size=2.0), True, False, False)[0]
        assert float(np.sqrt(np.mean((got.astype(np.float64) - want) ** 2))) <= 1e-5


@pytest.mark.gpu
def test_gpu_six_lanes_side_by_side():
    """batch.MAX_LANES device handles at once, asked for more: the correction tails' resident grids of six
    pairs fit the chip side by side, so nothing waits for a workgroup that cannot
    start; a bounded wait that expired would surface as MgxError.  Results bit-identical to single runs,
    three batches in a row."""
    from matchering_amd import stages

    cfg = mg.Config(max_piece_size=3.0)
    pairs = []
    for b in range(12):
        t = (0.5 * synth(12.0 + 0.5 * (b % 3), 44100, 1 + 2 * (b % 4))).astype(np.float32)
        r = np.clip(2.5 * synth(9.0, 44100, 2 + 2 * (b % 4)), -1, 1).astype(np.float32)
        pairs.append((t, r))
    single = [stages.main(t, r, cfg, need_default=True) for t, r in pairs[:4]]
    for _ in range(3):
        many = batch.master_many(pairs, cfg, need_default=True, lanes=16)
        for b, triple in enumerate(many):
            if b < 4:
                assert np.array_equal(triple[0], single[b][0])
            assert np.isfinite(triple[0]).all() and np.abs(triple[0]).max() <= 1.0


@pytest.mark.gpu
def test_bench_starts_its_own_ranks_and_rccl_sees_them():
    """``python bench.py --gpus 2`` with no launcher in front of it, on whatever GPUs
    this box has: the script starts two rank processes, they meet over matchering_amd.ranks, master their own pairs
    between two barriers, all-gather the FIR tables over RCCL and rank 0 prints ONE line.  On a one-GPU box the
    ranks share the GPU and RCCL runs over its loop-back socket transport (bench.share_one_gpu_over_rccl)."""
    import json
    import subprocess
    import sys

    from conftest import ROOT

    clean = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT")}
    run = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1",
                          "--workload", "8min_fir_only", "--no-traffic", "--no-gpu-state", "--no-cpu-baseline",
                          "--no-secondary", "--spinup", "0.05"], capture_output=True, text=True, timeout=600, env=clean)
    assert run.returncode == 0, run.stderr[-2000:]
    lines = run.stdout.splitlines()
    assert len(lines) == 1, run.stdout[:2000]                    # the line and nothing else (RCCL's banner goes to stderr)
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and line["launch"] == "self" and line["value"] > 0
    assert len(line["rank_seconds"]) == 2
    exchange = line["rccl_fir_allgather"]
    assert exchange["ok"] and exchange["ranks_seen"] == 2 and exchange["ranks_ok"] == 2 and exchange["distinct_tables"] == 2
    assert line["roofline"]["frac"] > 0.0
