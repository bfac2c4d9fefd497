"""Rendezvous of the rank processes of one node: barrier, max, broadcast and gather of small payloads.

The reference masters one pair per ``process`` call and has no notion of ranks (core.py:32-121); pairs share
nothing, so the only things rank processes ever tell each other are the 128-byte RCCL id (album mode's FIR
broadcast, bench.py's all-gather), "I am here" (the barriers around a timed region) and one number (the
slowest rank's time).  That does not need a collective library: rank 0 listens on a local stream socket, the
others connect, and every operation is one message to rank 0 and one back (a star; 8 ranks at most per node).

The socket is an abstract-namespace Unix socket named after MASTER_ADDR:MASTER_PORT -- the launcher's own
store (``torch.distributed.run`` keeps a TCP store on that very port) is left alone, nothing is left behind in
the file system, and two jobs with different ports do not meet.  ``MGX_RENDEZVOUS=tcp://host:port`` selects a
TCP socket instead (ranks on several nodes).  Every wait is bounded (``timeout`` seconds).

Messages are JSON (None, numbers, strings, lists, dicts; ``bytes`` as base64 under a reserved key), length-prefixed
and bounded -- never pickle: the abstract socket has no file permissions and the TCP form listens on the network, so
whatever connects must not be able to make a rank execute anything.  A rank introduces itself with its number and
``MGX_RENDEZVOUS_TOKEN`` (when the job sets one, rank 0 turns away connections that do not know it); numbers outside
1 .. world-1 and duplicates are refused.
"""

import base64
import hmac
import json
import os
import socket
import struct
import time

MAX_MESSAGE = 64 << 20          # bytes; the largest real payload is a gathered list of small dicts
_BYTES_KEY = "__mgx_bytes__"


def rank_environment():
    """(rank, world, local_rank) as a launcher exports them (RANK, WORLD_SIZE, LOCAL_RANK); (0, 1, 0) alone."""
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", str(rank)))
    if not 0 <= rank < world:
        raise ValueError(f"rank {rank} outside world of {world}")
    return rank, world, local


def ranks_are_on_this_node():
    """True unless ``MGX_RENDEZVOUS=tcp://host:port`` names another machine."""
    explicit = os.environ.get("MGX_RENDEZVOUS", "")
    if not explicit.startswith("tcp://"):
        return True
    host = explicit[6:].rpartition(":")[0]
    return host in ("", "127.0.0.1", "localhost", "::1")


def single_node_rccl_defaults():
    """Tell RCCL, before its first initialisation, that the job's ranks are the GPUs of ONE node.

    The data path between them is xGMI and the bootstrap needs no interface but loop-back, so neither is left to RCCL's
    probing of a box whose host name may not resolve: ``NCCL_SOCKET_IFNAME=lo`` and ``NCCL_IB_DISABLE=1``, each only
    where the host has not chosen itself, and not at all when the rendezvous names another machine (a loop-back
    bootstrap cannot reach it).  RCCL's first initialisation normally takes 2 - 5 s here either way
    (tools/rccl_init_time.py); on one box of the pool it took 457 s with RCCL's own defaults
    (profiles/r05_w_*).  This lives in the Python front end, not in libmgx: changing the environment of a host
    process is the host's decision -- a C/C++ host that binds ``mgx_comm_*`` sets the two
    variables itself -- and it runs where a job sets itself up (``Ranks()``, ``Device.comm_init``),
    which is before the batch lanes' threads exist.  Returns what it set."""
    if not ranks_are_on_this_node():
        return {}
    done = {}
    for key, value in (("NCCL_SOCKET_IFNAME", "lo"), ("NCCL_IB_DISABLE", "1")):
        if key not in os.environ:
            os.environ[key] = value
            done[key] = value
    return done


def _address():
    explicit = os.environ.get("MGX_RENDEZVOUS", "")
    if explicit.startswith("tcp://"):
        host, _, port = explicit[6:].rpartition(":")
        return socket.AF_INET, (host or "127.0.0.1", int(port))
    key = f"{os.environ.get('MASTER_ADDR', '127.0.0.1')}:{os.environ.get('MASTER_PORT', '29511')}"
    return socket.AF_UNIX, "\0mgx-ranks-" + key


def _encode(obj):
    if isinstance(obj, (bytes, bytearray, memoryview)):
        return {_BYTES_KEY: base64.b64encode(bytes(obj)).decode("ascii")}
    if isinstance(obj, (list, tuple)):
        return [_encode(v) for v in obj]
    if isinstance(obj, dict):
        return {str(k): _encode(v) for k, v in obj.items()}
    if obj is None or isinstance(obj, (bool, int, float, str)):
        return obj
    if hasattr(obj, "item"):                  # numpy scalars
        return obj.item()
    raise TypeError(f"ranks exchange None, numbers, strings, bytes, lists and dicts, not {type(obj).__name__}")


def _decode(obj):
    if isinstance(obj, dict):
        if set(obj) == {_BYTES_KEY}:
            return base64.b64decode(obj[_BYTES_KEY])
        return {k: _decode(v) for k, v in obj.items()}
    if isinstance(obj, list):
        return [_decode(v) for v in obj]
    return obj


def _send(sock, obj):
    blob = json.dumps(_encode(obj), allow_nan=True).encode("utf-8")
    if len(blob) > MAX_MESSAGE:
        raise ValueError(f"message of {len(blob)} bytes exceeds the {MAX_MESSAGE}-byte bound of the rendezvous")
    sock.sendall(struct.pack("<Q", len(blob)) + blob)


def _recv(sock):
    def exactly(n):
        parts = []
        while n:
            chunk = sock.recv(n)
            if not chunk:
                raise ConnectionError("a rank closed its connection")
            parts.append(chunk)
            n -= len(chunk)
        return b"".join(parts)

    (size,) = struct.unpack("<Q", exactly(8))
    if size > MAX_MESSAGE:
        raise ConnectionError(f"a peer announced a message of {size} bytes (bound: {MAX_MESSAGE})")
    return _decode(json.loads(exactly(size).decode("utf-8")))


def _token():
    return os.environ.get("MGX_RENDEZVOUS_TOKEN", "")


class Ranks:
    """The rank processes of one job.  With one rank every operation returns at once."""

    def __init__(self, rank=None, world=None, local=None, timeout=600.0):
        env = rank_environment()
        self.rank = env[0] if rank is None else int(rank)
        self.world = env[1] if world is None else int(world)
        self.local = env[2] if local is None else int(local)
        self.timeout = timeout
        self.peers = {}                      # rank 0: rank -> socket
        self.root = None                     # other ranks: the socket to rank 0
        self.listener = None
        if self.world > 1:
            single_node_rccl_defaults()
            self._meet()

    # ---- rendezvous ------------------------------------------------------------------------------------
    def _meet(self):
        family, address = _address()
        deadline = time.monotonic() + self.timeout
        if self.rank == 0:
            self.listener = socket.socket(family, socket.SOCK_STREAM)
            if family == socket.AF_INET:
                self.listener.setsockopt(socket.SOL_SOCKET, socket.SO_REUSEADDR, 1)
            self.listener.bind(address)
            self.listener.listen(self.world)
            while len(self.peers) < self.world - 1:
                self.listener.settimeout(max(0.1, deadline - time.monotonic()))
                try:
                    conn, _ = self.listener.accept()
                except socket.timeout:
                    missing = sorted(set(range(1, self.world)) - set(self.peers))
                    raise TimeoutError(f"ranks {missing} did not arrive within {self.timeout:.0f} s") from None
                conn.settimeout(self.timeout)
                try:                                   # whoever connects says who it is; anything else is turned away
                    hello = _recv(conn)
                    who = hello.get("rank") if isinstance(hello, dict) else None
                    known = isinstance(hello, dict) and hmac.compare_digest(str(hello.get("token", "")), _token())
                    if not (known and isinstance(who, int) and not isinstance(who, bool)
                            and 1 <= who < self.world and who not in self.peers):
                        raise ValueError(f"unexpected introduction {hello!r}"[:120])
                except (ValueError, ConnectionError, UnicodeDecodeError, socket.timeout):
                    conn.close()
                    continue
                self.peers[who] = conn
            for conn in self.peers.values():
                _send(conn, "met")
        else:
            while True:
                sock = socket.socket(family, socket.SOCK_STREAM)
                try:
                    sock.connect(address)
                    break
                except (ConnectionRefusedError, FileNotFoundError):
                    sock.close()
                    if time.monotonic() > deadline:
                        raise TimeoutError(f"rank 0 did not open the rendezvous within {self.timeout:.0f} s") from None
                    time.sleep(0.02)
            sock.settimeout(self.timeout)
            _send(sock, {"rank": self.rank, "token": _token()})
            self.root = sock
            _recv(sock)

    # ---- one round trip through rank 0 --------------------------------------------------------------------
    def _exchange(self, value, reduce):
        """Every rank contributes ``value``; ``reduce(list indexed by rank)`` runs on rank 0 and its result
        is what every rank returns."""
        if self.world == 1:
            return reduce([value])
        if self.rank == 0:
            values = [value] + [None] * (self.world - 1)
            for r, conn in self.peers.items():
                values[r] = _recv(conn)
            out = reduce(values)
            for conn in self.peers.values():
                _send(conn, out)
            return out
        _send(self.root, value)
        return _recv(self.root)

    def barrier(self):
        self._exchange(None, lambda values: None)

    def max(self, value):
        return float(self._exchange(float(value), max))

    def gather(self, value):
        """The values of all ranks, in rank order, on every rank."""
        return self._exchange(value, list)

    def broadcast_bytes(self, payload, size, root=0):
        """``payload`` of rank ``root`` (``size`` bytes) on every rank."""
        out = self._exchange(bytes(payload) if self.rank == root else None, lambda values: values[root])
        if len(out) != size:
            raise ValueError(f"broadcast payload has {len(out)} bytes, expected {size}")
        return out

    def finish(self):
        # (no closing barrier: every operation is a round trip through rank 0 that a rank completes by reading its
        # reply, so a rank that closes after its last operation leaves nobody waiting -- and rank 0 may go on
        # working for minutes, bench.py's CPU baseline, without the others holding on)
        for conn in list(self.peers.values()) + [self.root, self.listener]:
            if conn is not None:
                try:
                    conn.close()
                except OSError:
                    pass
        self.peers, self.root, self.listener = {}, None, None
