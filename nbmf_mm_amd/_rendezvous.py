"""Rendezvous for the multi-GPU fit: a process group built on the standard library only.

One process per GPU (ranks of ONE node, the xGMI domain).  What the ranks exchange on the host is tiny --
the 128-byte RCCL id or the HIP-IPC handle blocks at attach time, agreement flags, a max over ranks of a
timing -- so the group is a star over one listening socket owned by rank 0: every collective is "each rank
sends one message, rank 0's relay thread answers everybody with the list of all of them", and the others
(broadcast, barrier, reductions) are derived from that all-gather.  Sums are formed in rank order on every
rank, so an all-reduce is bitwise identical everywhere.  No PyTorch, no MPI.

Addressing (environment, as set by PyTorch's distributed launcher, ``bench.py``'s own launcher, or by
hand): ``RANK``, ``WORLD_SIZE``, ``MASTER_ADDR``, ``MASTER_PORT``.
  * default: an abstract Unix socket named after MASTER_PORT -- single node, nothing to clean up, and it
    cannot collide with the launcher's own store, which holds the TCP port itself under PyTorch's launcher;
  * ``NBMF_RDZV_PORT=<port>``: TCP on MASTER_ADDR:<port> instead (ranks that do not share a kernel).
Messages are pickled Python objects between processes of one job (the same trust domain as the launcher's).
"""
from __future__ import annotations

import os
import pickle
import socket
import struct
import threading
import time

import numpy as np

__all__ = ["Group", "SingleGroup", "init_from_env", "free_port"]

_HDR = struct.Struct("<Q")


def free_port() -> int:
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _send(sock, obj):
    data = pickle.dumps(obj, protocol=pickle.HIGHEST_PROTOCOL)
    sock.sendall(_HDR.pack(len(data)) + data)


def _recv_exact(sock, n):
    buf = bytearray(n)
    view, got = memoryview(buf), 0
    while got < n:
        r = sock.recv_into(view[got:], n - got)
        if r == 0:
            raise ConnectionError("rendezvous peer closed the connection")
        got += r
    return bytes(buf)


def _recv(sock):
    (n,) = _HDR.unpack(_recv_exact(sock, _HDR.size))
    return pickle.loads(_recv_exact(sock, n))


class SingleGroup:
    """The one-rank group (every collective is the identity): lets single-GPU callers use the sharded entry points."""

    world, rank = 1, 0

    def all_gather(self, obj):
        return [obj]

    def broadcast(self, obj, src=0):
        return obj

    def barrier(self):
        pass

    def all_reduce(self, arr, op="sum"):
        return arr

    def agree(self, ok):
        return bool(ok)

    def max_float(self, x):
        return float(x)

    def close(self):
        pass


class Group:
    """``world`` ranks joined through rank 0's relay.  Every rank must make the same sequence of calls."""

    def __init__(self, rank, world, address, timeout=300.0):
        """address: ``("unix", name)`` (abstract socket) or ``("tcp", host, port)``."""
        if world < 1 or not (0 <= rank < world):
            raise ValueError(f"bad rank {rank} / world {world}")
        self.rank, self.world, self._timeout = int(rank), int(world), float(timeout)
        self._relay = None
        self._listener = None
        self._relay_error = None
        if rank == 0:
            self._listener = self._listen(address)
            self._relay = threading.Thread(target=self._serve, name="nbmf-rendezvous", daemon=True)
            self._relay.start()
        self._sock = self._connect(address)
        _send(self._sock, ("hello", self.rank, self.world))
        ack = _recv(self._sock)
        if ack != ("welcome", self.world):
            raise ConnectionError(f"rendezvous handshake failed: {ack!r}")

    # ---- sockets ------------------------------------------------------------------------------------
    @staticmethod
    def _family(address):
        return socket.AF_UNIX if address[0] == "unix" else socket.AF_INET

    @staticmethod
    def _target(address):
        return "\0" + address[1] if address[0] == "unix" else (address[1], int(address[2]))

    def _listen(self, address):
        s = socket.socket(self._family(address), socket.SOCK_STREAM)
        if address[0] == "tcp":
            s.setsockopt(socket.SOL_SOCKET, socket.SO_REUSEADDR, 1)
        s.bind(self._target(address))
        s.listen(self.world)
        s.settimeout(self._timeout)
        return s

    def _connect(self, address):
        deadline = time.monotonic() + self._timeout
        while True:
            s = socket.socket(self._family(address), socket.SOCK_STREAM)
            try:
                s.connect(self._target(address))
                s.settimeout(self._timeout)
                if address[0] == "tcp":
                    s.setsockopt(socket.IPPROTO_TCP, socket.TCP_NODELAY, 1)
                return s
            except (ConnectionRefusedError, FileNotFoundError, socket.timeout):
                s.close()
                if time.monotonic() > deadline:
                    raise ConnectionError(f"rank {self.rank}: no rendezvous listener at {address} after {self._timeout:.0f} s")
                time.sleep(0.02)

    # ---- rank 0's relay: one round = one message from every rank, answered with the list of all -------
    def _serve(self):
        conns = [None] * self.world
        try:
            while any(c is None for c in conns):
                c, _ = self._listener.accept()
                c.settimeout(self._timeout)
                tag, r, w = _recv(c)
                if tag != "hello" or w != self.world or not (0 <= r < self.world) or conns[r] is not None:
                    _send(c, ("refused", f"unexpected hello {tag!r} rank {r} world {w}"))
                    c.close()
                    continue
                conns[r] = c
            for c in conns:
                _send(c, ("welcome", self.world))
            while True:
                msgs = []
                for c in conns:
                    c.settimeout(None if not msgs else self._timeout)   # idle between collectives is not an error
                    msgs.append(_recv(c))
                if any(tag == "bye" for tag, _ in msgs):     # envelopes: ("msg", payload) or ("bye", None)
                    break
                data = pickle.dumps([payload for _, payload in msgs], protocol=pickle.HIGHEST_PROTOCOL)
                frame = _HDR.pack(len(data)) + data
                for c in conns:
                    c.sendall(frame)
        except Exception as e:       # a rank died or timed out: the others see their sockets close
            self._relay_error = e
        finally:
            for c in conns:
                if c is not None:
                    try:
                        c.close()
                    except OSError:
                        pass
            self._listener.close()

    # ---- collectives --------------------------------------------------------------------------------
    def all_gather(self, obj):
        """List of every rank's ``obj``, in rank order, on every rank."""
        _send(self._sock, ("msg", obj))
        return _recv(self._sock)

    def broadcast(self, obj, src=0):
        return self.all_gather(obj if self.rank == src else None)[src]

    def barrier(self):
        self.all_gather(None)

    def all_reduce(self, arr, op="sum"):
        """In-place reduction of a NumPy array over the ranks (``sum``, ``min`` or ``max``), formed in rank order
        on every rank.  For small arrays and the host transport of the tests, not a data path."""
        parts = self.all_gather(np.ascontiguousarray(arr))
        acc = np.array(parts[0], copy=True)
        f = {"sum": np.add, "min": np.minimum, "max": np.maximum}[op]
        for p in parts[1:]:
            f(acc, p, out=acc)
        arr[...] = acc
        return arr

    def agree(self, ok):
        """True iff ``ok`` is true on every rank."""
        return all(self.all_gather(bool(ok)))

    def max_float(self, x):
        return max(self.all_gather(float(x)))

    def close(self):
        if self._sock is None:
            return
        try:
            _send(self._sock, ("bye", None))
        except OSError:
            pass
        if self._relay is not None:
            self._relay.join(5.0)
        try:
            self._sock.close()
        except OSError:
            pass
        self._sock = None

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()


def init_from_env(timeout=300.0):
    """The group described by RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT (see the module docstring)."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    if world == 1:
        return SingleGroup()
    host = os.environ.get("MASTER_ADDR", "127.0.0.1")
    port = os.environ.get("MASTER_PORT", "29500")
    if os.environ.get("NBMF_RDZV_PORT"):
        address = ("tcp", host, int(os.environ["NBMF_RDZV_PORT"]))
    else:
        # restarts by an elastic launcher get a fresh name: a relay of the previous attempt may still be closing
        address = ("unix", f"nbmf-rdzv-{port}-{os.environ.get('TORCHELASTIC_RESTART_COUNT', '0')}")
    return Group(rank, world, address, timeout=timeout)
