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

Who may join.  Nothing a peer sends is interpreted before the peer has proved that it belongs to the job:
  * Unix socket: both ends read the other's credentials from the kernel (``SO_PEERCRED``) and require the same
    user id -- an abstract socket has no file permissions and a guessable name, so another local user could
    otherwise connect to the relay first, or bind the name first and play relay;
  * a shared secret (``NBMF_RDZV_SECRET``, any string; ``bench.py``'s launcher draws a random one per job):
    mutual challenge-response with HMAC-SHA256 over fresh nonces, fixed-size frames.  REQUIRED for TCP, optional
    on top of the credential check for the Unix socket.
Messages are a small tagged binary encoding of None / bool / int / float / str / bytes / list / tuple / dict /
NumPy array -- decoding never executes anything (no pickle).
"""
from __future__ import annotations

import hmac
import os
import secrets
import socket
import struct
import threading
import time

import numpy as np

__all__ = ["Group", "SingleGroup", "init_from_env", "free_port", "encode", "decode"]

_HDR = struct.Struct("<Q")
_MAX_FRAME = 1 << 34          # 16 GiB: far above anything a job sends, far below a length field gone wrong
_MAGIC = b"NBMFRDZ2"


def free_port() -> int:
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


# ---- wire format -----------------------------------------------------------------------------------------
# one byte of tag, then: N none | T/F bool | i int64 | I big int (length-prefixed decimal) | d float64 |
# s str / b bytes (u64 length + data) | l list / t tuple (u64 count + items) | m dict (u64 count + key, value
# pairs) | a ndarray (dtype string, u8 ndim, u64 dims, raw C-order bytes; object dtypes are refused)
def _enc(obj, out):
    if obj is None:
        out.append(b"N")
    elif isinstance(obj, (bool, np.bool_)):
        out.append(b"T" if obj else b"F")
    elif isinstance(obj, (int, np.integer)):
        v = int(obj)
        if -(1 << 63) <= v < (1 << 63):
            out.append(b"i" + struct.pack("<q", v))
        else:
            s = str(v).encode()
            out.append(b"I" + _HDR.pack(len(s)) + s)
    elif isinstance(obj, (float, np.floating)):
        out.append(b"d" + struct.pack("<d", float(obj)))
    elif isinstance(obj, str):
        s = obj.encode("utf-8")
        out.append(b"s" + _HDR.pack(len(s)) + s)
    elif isinstance(obj, (bytes, bytearray, memoryview)):
        s = bytes(obj)
        out.append(b"b" + _HDR.pack(len(s)) + s)
    elif isinstance(obj, (list, tuple)):
        out.append((b"l" if isinstance(obj, list) else b"t") + _HDR.pack(len(obj)))
        for x in obj:
            _enc(x, out)
    elif isinstance(obj, dict):
        out.append(b"m" + _HDR.pack(len(obj)))
        for k, v in obj.items():
            _enc(k, out)
            _enc(v, out)
    elif isinstance(obj, np.ndarray):
        if obj.dtype.hasobject:
            raise TypeError("object arrays cannot cross the rendezvous")
        dt = obj.dtype.str.encode()
        out.append(b"a" + struct.pack("<B", len(dt)) + dt + struct.pack("<B", obj.ndim) +
                   b"".join(_HDR.pack(d) for d in obj.shape))
        out.append(np.ascontiguousarray(obj).tobytes())
    else:
        raise TypeError(f"cannot send {type(obj).__name__} through the rendezvous")


def encode(obj) -> bytes:
    out = []
    _enc(obj, out)
    return b"".join(out)


def _dec(buf, pos):
    tag = bytes(buf[pos:pos + 1])
    pos += 1
    if tag == b"N":
        return None, pos
    if tag == b"T":
        return True, pos
    if tag == b"F":
        return False, pos
    if tag == b"i":
        return struct.unpack_from("<q", buf, pos)[0], pos + 8
    if tag == b"d":
        return struct.unpack_from("<d", buf, pos)[0], pos + 8
    if tag in (b"I", b"s", b"b"):
        (n,) = _HDR.unpack_from(buf, pos)
        pos += 8
        if n > len(buf) - pos:
            raise ValueError("truncated rendezvous message")
        raw = bytes(buf[pos:pos + n])
        pos += n
        return (int(raw.decode()) if tag == b"I" else raw.decode("utf-8") if tag == b"s" else raw), pos
    if tag in (b"l", b"t"):
        (n,) = _HDR.unpack_from(buf, pos)
        pos += 8
        if n > len(buf) - pos:          # every item takes at least one byte
            raise ValueError("truncated rendezvous message")
        items = []
        for _ in range(n):
            x, pos = _dec(buf, pos)
            items.append(x)
        return (items if tag == b"l" else tuple(items)), pos
    if tag == b"m":
        (n,) = _HDR.unpack_from(buf, pos)
        pos += 8
        if 2 * n > len(buf) - pos:
            raise ValueError("truncated rendezvous message")
        d = {}
        for _ in range(n):
            k, pos = _dec(buf, pos)
            v, pos = _dec(buf, pos)
            d[k] = v
        return d, pos
    if tag == b"a":
        ln = buf[pos]
        pos += 1
        dt = np.dtype(bytes(buf[pos:pos + ln]).decode())
        pos += ln
        if dt.hasobject:
            raise ValueError("object arrays cannot cross the rendezvous")
        nd = buf[pos]
        pos += 1
        shape = struct.unpack_from("<%dQ" % nd, buf, pos)
        pos += 8 * nd
        count = 1
        for d in shape:
            count *= d
        nbytes = count * dt.itemsize
        if nbytes > len(buf) - pos:
            raise ValueError("truncated rendezvous message")
        arr = np.frombuffer(buf, dtype=dt, count=count, offset=pos).reshape(shape).copy()
        return arr, pos + nbytes
    raise ValueError(f"unknown tag {tag!r} in rendezvous message")


def decode(data):
    obj, pos = _dec(memoryview(data) if not isinstance(data, memoryview) else data, 0)
    if pos != len(data):
        raise ValueError("trailing bytes in rendezvous message")
    return obj


def _send(sock, obj):
    data = encode(obj)
    sock.sendall(_HDR.pack(len(data)) + data)


def _recv_exact(sock, n):
    buf = bytearray(n)
    view, got = memoryview(buf), 0
    while got < n:
        r = sock.recv_into(view[got:], n - got)
        if r == 0:
            raise ConnectionError("rendezvous peer closed the connection")
        got += r
    return buf


def _recv(sock):
    (n,) = _HDR.unpack(_recv_exact(sock, _HDR.size))
    if n > _MAX_FRAME:
        raise ConnectionError(f"rendezvous frame of {n} bytes refused")
    return decode(_recv_exact(sock, n))


# ---- admission ---------------------------------------------------------------------------------------------
def _same_user(sock):
    """Unix socket: the peer's user id as the kernel recorded it at connect()/listen() time."""
    cred = sock.getsockopt(socket.SOL_SOCKET, socket.SO_PEERCRED, struct.calcsize("3i"))
    _pid, uid, _gid = struct.unpack("3i", cred)
    return uid == os.getuid()


def _mac(key, role, n_cli, n_srv):
    return hmac.new(key, role + n_cli + n_srv, "sha256").digest()


def _handshake_client(sock, key):
    n_cli = secrets.token_bytes(16)
    sock.sendall(_MAGIC + n_cli)
    reply = bytes(_recv_exact(sock, 16 + 32))
    n_srv, tag = reply[:16], reply[16:]
    if not hmac.compare_digest(tag, _mac(key, b"srv", n_cli, n_srv)):
        raise ConnectionError("rendezvous listener failed authentication (NBMF_RDZV_SECRET differs, or it is not this job's relay)")
    sock.sendall(_mac(key, b"cli", n_cli, n_srv))


def _handshake_server(sock, key):
    hello = bytes(_recv_exact(sock, len(_MAGIC) + 16))
    if hello[:len(_MAGIC)] != _MAGIC:
        return False
    n_cli, n_srv = hello[len(_MAGIC):], secrets.token_bytes(16)
    sock.sendall(n_srv + _mac(key, b"srv", n_cli, n_srv))
    return hmac.compare_digest(bytes(_recv_exact(sock, 32)), _mac(key, b"cli", n_cli, n_srv))


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

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()


class LocalGroup:
    """``world`` ranks that are THREADS of one process (``NBMF(n_gpus=N)``: one host thread, context and stream per GPU):
    the same small interface as :class:`Group`, over a barrier and a shared table instead of sockets.  ``make(world)``
    returns the ranks' group objects.  A rank that fails calls ``abort()`` (its ``__exit__`` does, on an exception): the
    others' collectives then raise ``ConnectionError`` instead of waiting for it.  A rank that never ARRIVES -- stuck in
    a HIP call: a hung kernel, a collective whose partner is gone -- is given ``timeout`` seconds (default
    ``NBMF_LOCAL_GROUP_TIMEOUT_S`` or 1800: long enough for the slowest rank's upload of a big shard, many times the
    peer transport's own bound); then every waiting rank raises a ``ConnectionError`` that names the missing ranks."""

    DEFAULT_TIMEOUT_S = 1800.0

    class _Shared:
        def __init__(self, world, timeout):
            import threading
            self.barrier = threading.Barrier(world)
            self.slots = [None] * world
            self.timeout = timeout
            self.arrivals = [0] * world        # how many barrier waits each rank has entered (diagnosis of a timeout)
            self.timed_out = None              # (missing ranks, timeout) once a wait has run out: what every later failure reports

    def __init__(self, shared, rank, world):
        self._s, self.rank, self.world = shared, rank, world

    @classmethod
    def make(cls, world, timeout=None):
        import os
        if timeout is None:
            timeout = float(os.environ.get("NBMF_LOCAL_GROUP_TIMEOUT_S", cls.DEFAULT_TIMEOUT_S))
        shared = cls._Shared(int(world), timeout)
        return [cls(shared, r, int(world)) for r in range(int(world))]

    def _wait(self):
        import threading
        import time
        s = self._s
        s.arrivals[self.rank] += 1
        mine, t0 = s.arrivals[self.rank], time.monotonic()
        try:
            s.barrier.wait(s.timeout)
        except threading.BrokenBarrierError:
            # broken by a failed rank's abort(), or by a timeout -- this rank's own or, once that has broken the barrier,
            # somebody else's: whoever has not entered this wait is the one everybody was waiting for
            missing = [r for r, n in enumerate(s.arrivals) if n < mine]
            if missing and s.timeout is not None and time.monotonic() - t0 >= 0.9 * s.timeout and s.timed_out is None:
                s.timed_out = (missing, s.timeout)
            # the diagnosis belongs to the group, not to whoever happened to wait longest: a rank that enters a collective
            # AFTER the barrier broke (the late one itself, say) reports the same ranks and the same bound
            if s.timed_out is not None:
                gone, bound = s.timed_out
                raise ConnectionError(f"rank(s) {gone} of this process did not reach a collective within {bound:g} s "
                                      f"(stuck in a device call?)") from None
            raise ConnectionError("another rank of this process has failed") from None

    def all_gather(self, obj):
        self._s.slots[self.rank] = obj
        self._wait()                       # everybody has written
        out = list(self._s.slots)
        self._wait()                       # everybody has read: the table may be written again
        return out

    def broadcast(self, obj, src=0):
        return self.all_gather(obj if self.rank == src else None)[src]

    def barrier(self):
        self._wait()

    def all_reduce(self, arr, op="sum"):
        parts = self.all_gather(np.array(arr, copy=True))
        acc = np.array(parts[0], copy=True)
        f = {"sum": np.add, "min": np.minimum, "max": np.maximum}[op]
        for p in parts[1:]:                # rank order on every rank: identical bits
            f(acc, p, out=acc)
        arr[...] = acc
        return arr

    def agree(self, ok):
        return all(self.all_gather(bool(ok)))

    def max_float(self, x):
        return max(self.all_gather(float(x)))

    def abort(self):
        self._s.barrier.abort()

    def close(self):
        pass

    def __enter__(self):
        return self

    def __exit__(self, exc_type, *exc):
        if exc_type is not None:
            self.abort()


class Group:
    """``world`` ranks joined through rank 0's relay.  Every rank must make the same sequence of calls."""

    def __init__(self, rank, world, address, timeout=300.0, secret=None, collective_timeout=None):
        """address: ``("unix", name)`` (abstract socket) or ``("tcp", host, port)``.
        timeout: how long to wait for the listener / for all ranks to arrive / for a handshake step.
        collective_timeout: bound on the wait inside a collective once the group stands (None = no bound: ranks that
        do uneven amounts of work between collectives -- restarts, grid points -- may be minutes apart; a rank that
        DIES is still noticed at once, its socket closes).
        secret: bytes shared by the ranks of the job; default ``NBMF_RDZV_SECRET`` from the environment.  Required for TCP."""
        if world < 1 or not (0 <= rank < world):
            raise ValueError(f"bad rank {rank} / world {world}")
        if secret is None and os.environ.get("NBMF_RDZV_SECRET"):
            secret = os.environ["NBMF_RDZV_SECRET"].encode()
        if address[0] == "tcp" and not secret:
            raise ValueError("a TCP rendezvous needs a shared secret (NBMF_RDZV_SECRET, or secret=): anyone who can reach "
                             "the port could otherwise join the job")
        self._key = bytes(secret or b"")
        self._unix = address[0] == "unix"
        self.rank, self.world, self._timeout = int(rank), int(world), float(timeout)
        self._ctimeout = None if collective_timeout is None else float(collective_timeout)
        self._relay = None
        self._listener = None
        self._relay_error = None
        self._sock = None
        if rank == 0:
            self._listener = self._listen(address)
            self._relay = threading.Thread(target=self._serve, name="nbmf-rendezvous", daemon=True)
            self._relay.start()
        self._sock = self._connect(address)
        _send(self._sock, ("hello", self.rank, self.world))
        ack = _recv(self._sock)
        if ack != ("welcome", self.world):
            raise ConnectionError(f"rendezvous handshake failed: {ack!r}")
        self._sock.settimeout(self._ctimeout)

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
        s.listen(self.world + 8)
        s.settimeout(self._timeout)
        return s

    def _connect(self, address):
        deadline = time.monotonic() + self._timeout
        while True:
            s = socket.socket(self._family(address), socket.SOCK_STREAM)
            try:
                s.connect(self._target(address))
            except (ConnectionRefusedError, FileNotFoundError, socket.timeout):
                s.close()
                if time.monotonic() > deadline:
                    raise ConnectionError(f"rank {self.rank}: no rendezvous listener at {address} after {self._timeout:.0f} s")
                time.sleep(0.02)
                continue
            s.settimeout(self._timeout)
            if address[0] == "tcp":
                s.setsockopt(socket.IPPROTO_TCP, socket.TCP_NODELAY, 1)
                s.setsockopt(socket.SOL_SOCKET, socket.SO_KEEPALIVE, 1)
            elif not _same_user(s):
                s.close()
                raise ConnectionError(f"rank {self.rank}: the rendezvous listener at {address} belongs to another user")
            _handshake_client(s, self._key)
            return s

    # ---- rank 0's relay: one round = one message from every rank, answered with the list of all -------
    def _admit(self, c):
        """Authenticate a fresh connection and read its hello; returns the rank or None (connection closed)."""
        try:
            c.settimeout(min(self._timeout, 30.0))
            if self._unix and not _same_user(c):
                raise ConnectionError("peer of another user")
            if not _handshake_server(c, self._key):
                raise ConnectionError("authentication failed")
            if not self._unix:
                c.setsockopt(socket.SOL_SOCKET, socket.SO_KEEPALIVE, 1)
            tag, r, w = _recv(c)
            if tag != "hello" or w != self.world or not isinstance(r, int) or not (0 <= r < self.world):
                _send(c, ("refused", f"unexpected hello {tag!r} rank {r} world {w}"))
                raise ConnectionError("bad hello")
            return r
        except Exception:            # a stranger, a port scanner, a rank of another job: not this job's problem
            try:
                c.close()
            except OSError:
                pass
            return None

    def _serve(self):
        conns = [None] * self.world
        lock = threading.Lock()

        def admit(c):             # one thread per fresh connection: a silent stranger stalls nobody but itself
            r = self._admit(c)
            if r is None:
                return
            with lock:
                taken = conns[r] is not None
                if not taken:
                    conns[r] = c
            if taken:
                try:
                    _send(c, ("refused", f"rank {r} is already here"))
                    c.close()
                except OSError:
                    pass

        try:
            deadline = time.monotonic() + self._timeout
            self._listener.settimeout(0.05)
            while True:
                with lock:
                    if all(c is not None for c in conns):
                        break
                if time.monotonic() > deadline:
                    raise ConnectionError(f"only {sum(c is not None for c in conns)} of {self.world} ranks arrived in {self._timeout:.0f} s")
                try:
                    c, _ = self._listener.accept()
                except socket.timeout:
                    continue
                threading.Thread(target=admit, args=(c,), daemon=True).start()
            for c in conns:
                c.settimeout(self._ctimeout)
                _send(c, ("welcome", self.world))
            while True:
                frames = []
                for c in conns:
                    (n,) = _HDR.unpack(_recv_exact(c, _HDR.size))
                    if n > _MAX_FRAME:
                        raise ConnectionError(f"frame of {n} bytes refused")
                    frames.append(bytes(_recv_exact(c, n)))
                # envelopes ("msg", payload) / ("bye", None): the relay looks at the tag only and passes the payload
                # bytes on undecoded -- b"t" + count(2) + b"s" + len + tag ...
                if any(not f.startswith(_MSG_PREFIX) for f in frames):
                    break            # a ("bye", None) envelope (or anything that is not a message): the group ends
                # Every frame must be a well-formed message before it is spliced into what ALL ranks receive: a
                # malformed one would otherwise surface as a decode error on every rank.  (Decoding executes nothing;
                # the relay still forwards the bytes it was given.)  A rank that sends garbage is treated as a rank
                # that died: the group ends, the others see their sockets close.
                for r, f in enumerate(frames):
                    try:
                        decode(f)
                    except Exception as e:
                        raise ConnectionError(f"rank {r} sent a malformed frame: {e}") from None
                body = b"l" + _HDR.pack(len(frames)) + b"".join(f[_ENV_SKIP:] for f in frames)
                out = _HDR.pack(len(body)) + body
                for c in conns:
                    c.sendall(out)
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


# bytes of an encoded ("msg", payload) envelope in front of the payload: tuple tag + count, str tag + length + "msg"
_MSG_PREFIX = encode(("msg", None))[:-1]
_ENV_SKIP = len(_MSG_PREFIX)


def init_from_env(timeout=300.0, collective_timeout=None, rank=None, world=None):
    """The group described by RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT (see the module docstring); ``rank`` /
    ``world`` override the environment (a process that hosts several ranks, one thread each, numbers them itself)."""
    world = int(os.environ.get("WORLD_SIZE", "1")) if world is None else int(world)
    rank = int(os.environ.get("RANK", "0")) if rank is None else int(rank)
    if world == 1:
        return SingleGroup()
    host = os.environ.get("MASTER_ADDR", "127.0.0.1")
    port = os.environ.get("MASTER_PORT", "29500")
    if os.environ.get("NBMF_RDZV_PORT"):
        address = ("tcp", host, int(os.environ["NBMF_RDZV_PORT"]))
    else:
        # restarts by an elastic launcher, and bench.py's own second attempt (NBMF_RDZV_GENERATION), get a fresh name: a relay
        # of the previous attempt may still be closing
        address = ("unix", f"nbmf-rdzv-{port}-{os.environ.get('TORCHELASTIC_RESTART_COUNT', '0')}"
                           f"-{os.environ.get('NBMF_RDZV_GENERATION', '0')}")
    return Group(rank, world, address, timeout=timeout, collective_timeout=collective_timeout)
