"""The N>1 path of bench.py on CPU: two processes over gloo (world_size 2).

Pairs are independent, so the multi-GPU path is: rank r masters pair r, no data-path collective, a
barrier + max-over-ranks around the timed region, and a broadcast of the 128-byte RCCL id.  Those
host-side pieces (bench.Ranks, bench.timed_steps, the per-rank synthetic pair) are exercised here
without a GPU; the RCCL all-gather of the FIR tables itself needs GPUs and runs in bench.py.
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

    ranks = bench.Ranks()
    try:
        assert ranks.rank == rank and ranks.world == world

        calls, syncs = [], []                   # the second argument stands in for the HIP stream synchronisation
        elapsed = bench.timed_steps(ranks, lambda: syncs.append(1), lambda: calls.append(1), steps=3, warmup=2)
        assert len(calls) == 5 and len(syncs) == 2 and elapsed >= 0.0
        slow = ranks.max(10.0 + rank)           # the job's time is the slowest rank's
        payload = bytes(range(128)) if rank == 0 else bytes(128)
        uid = ranks.broadcast_bytes(payload, 128)
        target, _ = make_pair(0.05, 44100, pair=ranks.rank)
        queue.put((rank, slow, uid == bytes(range(128)), float(np.abs(target).sum())))
    finally:
        ranks.finish()


def test_two_ranks_over_gloo():
    ctx = mp.get_context("spawn")
    queue = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, queue)) for r in range(2)]
    for p in procs:
        p.start()
    results = sorted(queue.get(timeout=300) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    (r0, max0, ok0, sum0), (r1, max1, ok1, sum1) = results
    assert (r0, r1) == (0, 1)
    assert max0 == max1 == 11.0                # max over ranks, identical everywhere
    assert ok0 and ok1                          # rank 0's 128-byte id reached rank 1
    assert sum0 != sum1                         # every rank masters its own pair
