"""How long the FIRST RCCL initialisation of a process takes (one-rank communicator on GPU 0), and a broadcast through it.

    python tools/rccl_init_time.py            # the library's single-node defaults (NCCL_SOCKET_IFNAME=lo, NCCL_IB_DISABLE=1)
    NCCL_SOCKET_IFNAME= NCCL_IB_DISABLE=0 python tools/rccl_init_time.py     # RCCL's own defaults
"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from matchering_amd.device import Device

dev = Device(0)
buf = dev.upload(np.arange(8192, dtype=np.float32))
dev.synchronize()
t0 = time.perf_counter()
dev.comm_init(0, 1)
t1 = time.perf_counter()
dev.comm_broadcast(buf, 8192, 0)
dev.synchronize()
t2 = time.perf_counter()
dev.comm_destroy()
print(f"rccl init {t1 - t0:.2f} s, first broadcast {t2 - t1:.3f} s, NCCL_SOCKET_IFNAME={os.environ.get('NCCL_SOCKET_IFNAME')!r} "
      f"NCCL_IB_DISABLE={os.environ.get('NCCL_IB_DISABLE')!r}")
dev.close()
