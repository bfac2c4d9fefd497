"""The N>1 path on CPU: real rank processes (world_size 2 and 8).

Pairs are independent, so the multi-GPU path is: rank r masters pair r, no data-path collective, a
barrier + max-over-ranks around the timed region, and a broadcast of the 128-byte RCCL id.  Those
host-side pieces (matchering_amd.ranks.Ranks -- a local socket, no torch --, bench.timed_steps, the
per-rank synthetic pair) are exercised here without a GPU; the RCCL all-gather of the FIR tables itself
needs GPUs and runs in bench.py.  The batch front end is run under a two-process ``gloo`` world as a
launcher-started job would be (``torch.distributed.run`` exports the same RANK / WORLD_SIZE / LOCAL_RANK).
"""

import multiprocessing as mp
import os
import socket
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, queue):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank),
                      MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    sys.path.insert(0, ROOT)
    import bench
    from matchering_amd.synth import make_pair

    ranks = bench.Ranks()                       # matchering_amd.ranks.Ranks: a local socket, no torch
    try:
        assert ranks.rank == rank and ranks.world == world

        calls, syncs = [], []                   # the second argument stands in for the HIP stream synchronisation
        elapsed = bench.timed_steps(ranks, lambda: syncs.append(1), lambda: calls.append(1), steps=3, warmup=2)
        assert len(calls) == 5 and len(syncs) == 2 and elapsed >= 0.0
        slow = ranks.max(10.0 + rank)           # the job's time is the slowest rank's
        payload = bytes(range(128)) if rank == 0 else bytes(128)
        uid = ranks.broadcast_bytes(payload, 128)
        everyone = ranks.gather({"rank": rank})
        target, _ = make_pair(0.05, 44100, pair=ranks.rank)
        queue.put((rank, slow, uid == bytes(range(128)), float(np.abs(target).sum()), [e["rank"] for e in everyone]))
    finally:
        ranks.finish()


def _run_world(world):
    ctx = mp.get_context("spawn")
    queue = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, queue)) for r in range(world)]
    for p in procs:
        p.start()
    results = sorted(queue.get(timeout=300) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    return results


def test_two_ranks_over_the_local_socket():
    (r0, max0, ok0, sum0, seen0), (r1, max1, ok1, sum1, seen1) = _run_world(2)
    assert (r0, r1) == (0, 1)
    assert max0 == max1 == 11.0                # max over ranks, identical everywhere
    assert ok0 and ok1                          # rank 0's 128-byte id reached rank 1
    assert sum0 != sum1                         # every rank masters its own pair
    assert seen0 == seen1 == [0, 1]             # gather: everyone, in rank order, everywhere


def test_eight_ranks_over_the_local_socket():
    results = _run_world(8)                     # a node's worth of ranks (the driver's --gpus 8)
    assert [r[0] for r in results] == list(range(8))
    assert all(r[1] == 17.0 and r[2] and r[4] == list(range(8)) for r in results)
    assert len({r[3] for r in results}) == 8


def test_a_missing_rank_is_a_timeout_not_a_hang():
    sys.path.insert(0, ROOT)
    from matchering_amd.ranks import Ranks

    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()))
    try:
        import pytest

        with pytest.raises(TimeoutError):
            Ranks(rank=0, world=2, local=0, timeout=0.5)          # rank 1 never comes
        with pytest.raises(TimeoutError):
            Ranks(rank=1, world=2, local=1, timeout=0.5)          # rank 0 never listens
    finally:
        os.environ.pop("MASTER_PORT", None)
        os.environ.pop("MASTER_ADDR", None)


def test_the_rendezvous_executes_nothing_a_peer_sends():
    """the socket has no permissions, so whatever connects may be hostile.  Messages are JSON (no
    pickle); a pickle payload, an out-of-range or duplicate rank number and a wrong token are turned away, and the
    real rank still gets in."""
    import pickle
    import struct
    import threading

    sys.path.insert(0, ROOT)
    from matchering_amd import ranks as R

    port = _free_port()
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), MGX_RENDEZVOUS_TOKEN="s3cret")
    try:
        family, address = R._address()
        result = {}

        def root():
            r0 = R.Ranks(rank=0, world=2, local=0, timeout=20.0)
            result["gathered"] = r0.gather({"rank": 0, "blob": b"\x00\xff"})
            r0.finish()

        t = threading.Thread(target=root)
        t.start()

        def knock(payload_bytes):
            for _ in range(200):
                s = socket.socket(family, socket.SOCK_STREAM)
                try:
                    s.connect(address)
                    break
                except (ConnectionRefusedError, FileNotFoundError):
                    s.close()
                    import time
                    time.sleep(0.02)
            s.sendall(struct.pack("<Q", len(payload_bytes)) + payload_bytes)
            s.settimeout(2.0)
            try:
                return s.recv(8)              # b"" = turned away
            except (socket.timeout, ConnectionError):
                return None
            finally:
                s.close()

        class Boom:
            def __reduce__(self):
                return (os.system, ("touch /tmp/mgx_ranks_pwned",))

        if os.path.exists("/tmp/mgx_ranks_pwned"):
            os.remove("/tmp/mgx_ranks_pwned")
        assert knock(pickle.dumps(Boom())) == b""                                        # not JSON
        assert knock(b'{"rank": 7, "token": "s3cret"}') == b""                            # outside the world
        assert knock(b'{"rank": 1, "token": "wrong"}') == b""                             # does not know the token
        assert knock(b'{"rank": true, "token": "s3cret"}') == b""                         # not a number
        assert not os.path.exists("/tmp/mgx_ranks_pwned")
        r1 = R.Ranks(rank=1, world=2, local=1, timeout=20.0)
        got = r1.gather({"rank": 1, "blob": b"\x01"})
        r1.finish()
        t.join(20.0)
        assert not t.is_alive()
        assert got == result["gathered"] == [{"rank": 0, "blob": b"\x00\xff"}, {"rank": 1, "blob": b"\x01"}]
    finally:
        for k in ("MASTER_PORT", "MASTER_ADDR", "MGX_RENDEZVOUS_TOKEN"):
            os.environ.pop(k, None)


# ---- the batch front end itself under a real two-process world ---------------
def _batch_worker(rank, world, port, folder, queue):
    """What ``python -m torch.distributed.run ... -m matchering_amd.batch jobs.json`` does on each rank, with
    the CPU oracle behind the ``master`` hook: ranks come from the launcher's environment, every rank masters
    its own share of the jobs and writes its own files; gloo is used for the rendezvous and two barriers only
    (the batch path has no data-path collective)."""
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank),
                      MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import torch.distributed as dist

    import matchering_amd as mg
    from matchering_amd import batch
    from test_batch import _oracle_master

    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        jobs = batch.jobs_from_json(os.path.join(folder, "jobs.json"))
        cfg = mg.Config(internal_sample_rate=8000, fft_size=256, max_piece_size=2)
        dist.barrier()
        mine = batch.process_batch(jobs, cfg, lanes=2, io_threads=2, master=_oracle_master)      # rank, world: from the environment
        dist.barrier()
        many = batch.master_many([(np.zeros((4, 2), np.float32),) * 2] * 3, cfg, lanes=None,
                                 master=lambda t, r, c, *needs: (rank, None, None))             # lanes=None without a GPU
        queue.put((rank, mine, [m[0] for m in many]))
    finally:
        dist.destroy_process_group()


def test_process_batch_in_two_rank_processes(tmp_path):
    import json

    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import mastering_oracle as mo
    from matchering_amd import audio_io
    from matchering_amd.synth import synth

    rate, jobs = 8000, []
    for b in range(5):
        t = (0.5 * synth(1.5, rate, 1 + 2 * b)).astype(np.float32)
        r = np.clip(2.5 * synth(1.5, rate, 2 + 2 * b), -1, 1).astype(np.float32)
        tp, rp = str(tmp_path / f"t{b}.wav"), str(tmp_path / f"r{b}.wav")
        audio_io.write_wav(tp, t, rate, "FLOAT")
        audio_io.write_wav(rp, r, rate, "FLOAT")
        jobs.append({"target": tp, "reference": rp, "results": [{"file": str(tmp_path / f"out{b}.wav"), "subtype": "FLOAT"}]})
    (tmp_path / "jobs.json").write_text(json.dumps(jobs))
    ctx = mp.get_context("spawn")
    queue = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_batch_worker, args=(r, 2, port, str(tmp_path), queue)) for r in range(2)]
    for p in procs:
        p.start()
    results = sorted(queue.get(timeout=300) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    (r0, mine0, tags0), (r1, mine1, tags1) = results
    assert (r0, mine0, r1, mine1) == (0, [0, 2, 4], 1, [1, 3])          # pair i -> rank i mod world, nothing twice
    assert tags0 == [0, 0, 0] and tags1 == [1, 1, 1]
    ocfg = mo.params(internal_sample_rate=rate, fft_size=256, max_piece_size=2)
    for b, job in enumerate(jobs):
        t, _ = audio_io.read_wav(job["target"])
        r, _ = audio_io.read_wav(job["reference"])
        want = mo.master(t.astype(np.float64), r.astype(np.float64), ocfg, True, False, False)[0]
        got, _ = audio_io.read_wav(job["results"][0]["file"])
        assert np.abs(got - want).max() <= 1e-6
