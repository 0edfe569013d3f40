"""bench.py's N > 1 machinery on the CPU: the supervisor that turns a failed attempt into ONE fresh retry over the host
transport (never a re-exec of a process that has used the GPU), the watchdog behind `setup_bound_s`, and the extra timed leg
that puts RCCL's own rate and rank count on the line whatever transport won (`_dist.time_transport`, with stand-in contexts)."""
import json
import os
import subprocess
import sys
import threading
import time

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FAKE = [sys.executable, os.path.join(ROOT, "tests", "fake_bench_worker.py")]


def _supervise(mode, argv, capfd, **kw):
    sys.path.insert(0, ROOT)
    import bench
    os.environ["FAKE_MODE"] = mode
    try:
        code = bench.supervise(argv, worker=FAKE, **kw)
    finally:
        del os.environ["FAKE_MODE"]
    cap = capfd.readouterr()
    lines = [json.loads(ln) for ln in cap.out.splitlines() if ln.startswith("{")]
    return code, lines, cap.err


def test_supervisor_passes_a_good_attempt_through(capfd):
    code, lines, _ = _supervise("ok", ["--gpus", "2", "--steps", "3"], capfd)
    assert code == 0 and lines == [{"attempt": 0, "argv": ["--gpus", "2", "--steps", "3"], "failed": None}]


def test_supervisor_retries_once_over_the_host_transport_and_hands_on_what_failed(capfd):
    code, lines, err = _supervise("fail_then_ok", ["--gpus", "2", "--transport", "auto"], capfd)
    assert code == 0 and len(lines) == 1
    assert lines[0]["attempt"] == 1 and lines[0]["argv"] == ["--gpus", "2", "--transport", "auto", "--transport", "host"]
    assert lines[0]["failed"] == "exit code 75: NBMFHipError: peer exchange timed out"
    assert "a fresh worker retries over the host transport" in err


def test_supervisor_kills_a_worker_that_never_ends_and_retries(capfd):
    t0 = time.monotonic()
    code, lines, _ = _supervise("hang_then_ok", ["--gpus", "2"], capfd, attempt_timeout=1.0)
    assert code == 0 and lines[0]["attempt"] == 1 and "last-resort limit of 1 s" in lines[0]["failed"]
    assert time.monotonic() - t0 < 30


def test_supervisor_retries_after_a_worker_that_died_of_a_signal(capfd):
    code, lines, _ = _supervise("crash_then_ok", ["--gpus", "2"], capfd)
    assert code == 0 and lines[0]["attempt"] == 1 and lines[0]["failed"].startswith("exit code -11")


def test_supervisor_gives_up_after_the_retry_and_never_retries_the_host_transport(capfd):
    code, lines, err = _supervise("always_fail", ["--gpus", "2"], capfd)
    assert code == 3 and lines == [] and "attempt 1 failed (exit code 3: attempt 1 broke); giving up" in err
    code, lines, err = _supervise("always_fail", ["--gpus", "2", "--transport", "host"], capfd)
    assert code == 3 and "attempt 0 failed" in err and "attempt 1" not in err


def test_watchdog_fires_once_past_its_bound_and_not_when_disarmed():
    sys.path.insert(0, ROOT)
    import bench
    fired = []
    dog = bench.Watchdog()
    dog.arm(0.2, "phase one", fired.append)
    dog.disarm()
    time.sleep(0.4)
    assert fired == []
    dog.arm(0.2, "phase two", fired.append)
    time.sleep(0.6)
    assert fired == ["phase two"] and dog.armed_for == 0.2


class _Ctx:
    """Stand-in context for _dist.time_transport: RCCL attaches unless told otherwise; iterations take `slow` seconds."""

    def __init__(self, rank, world, fail=(), slow=0.0, sees=None):
        self.rank, self.world, self.fail, self.slow, self.calls, self.attached = rank, world, set(fail), slow, [], None
        self.sees = world if sees is None else sees

    def comm_init(self, uid, world, rank, axis):
        from nbmf_mm_amd import _hip
        self.calls.append("rccl")
        if "attach" in self.fail:
            raise _hip.NBMFHipError("ncclCommInitRank failed: unhandled system error")
        self.attached = "rccl"

    def comm_detach(self):
        self.calls.append("detach")
        self.attached = None

    def comm_info(self):
        return {"kind": self.attached, "nranks_seen": self.sees, "remote": self.rank}

    def run(self, n, tol):
        from nbmf_mm_amd import _hip
        self.calls.append(f"run{n}")
        if "run" in self.fail and n > 2:
            raise _hip.NBMFHipError("ncclAllReduce failed")
        time.sleep(self.slow * n)

    def synchronize(self):
        pass


def test_time_transport_reports_rate_rank_counts_and_failures_identically_on_every_rank():
    """Three ranks (threads over LocalGroup) time RCCL beside whatever they run on: the rate is steps over the SLOWEST rank's
    time, `nranks_seen` lists what every rank's communicator itself reports (a rank whose communicator joins fewer ranks
    shows), an attach refused on one rank or an exchange failing on one rank gives every rank the same error and no rate,
    and the context is detached again in every case."""
    from nbmf_mm_amd import _dist, _hip, _rendezvous
    _hip_uid, _hip.comm_unique_id = _hip.comm_unique_id, lambda: bytes(128)
    try:
        def job(make):
            groups = _rendezvous.LocalGroup.make(3, timeout=20)
            out, ctxs = [None] * 3, [make(r) for r in range(3)]

            def body(r):
                resets = []
                out[r] = (_dist.time_transport(ctxs[r], groups[r], lambda: resets.append(1), "rccl", steps=4, warmup=2), len(resets))
            ts = [threading.Thread(target=body, args=(r,)) for r in range(3)]
            for t in ts:
                t.start()
            for t in ts:
                t.join(60)
            return out, ctxs
        out, ctxs = job(lambda r: _Ctx(r, 3, slow=0.05 if r == 2 else 0.0, sees=3 if r else 2))
        assert all(o == out[0] for o in out)
        rec, n_reset = out[0]
        assert n_reset == 1 and rec["error"] is None and rec["nranks_seen"] == [2, 3, 3] and rec["remote"] == [0, 1, 2]
        assert 4 / 0.5 < rec["value"] <= 4 / 0.2                       # four iterations at 0.05 s on the slowest rank
        assert all(c.calls == ["rccl", "run2", "run4", "detach"] and c.attached is None for c in ctxs)
        out, ctxs = job(lambda r: _Ctx(r, 3, fail=["attach"] if r == 1 else []))
        assert all(o[0] == out[0][0] for o in out) and out[0][0]["value"] is None and "did not attach" in out[0][0]["error"]
        assert all(c.attached is None for c in ctxs)
        out, ctxs = job(lambda r: _Ctx(r, 3, fail=["run"] if r == 0 else []))
        assert all(o[0] == out[0][0] for o in out) and out[0][0]["value"] is None
        assert out[0][0]["error"] == "rank 0: timed run over rccl failed: ncclAllReduce failed"
        assert all(c.calls[-1] == "detach" for c in ctxs)
    finally:
        _hip.comm_unique_id = _hip_uid
