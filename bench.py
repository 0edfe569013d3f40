#!/usr/bin/env python3
"""Headline benchmark: MM-iterations/sec of the NBMF-MM inner loop on MI355X.

Workload (BASELINE.json metric / configs[2]): synthetic dense binary V 65536 x 8192 handed over as
float64, K=64, mask with 90 % observed entries, alpha=beta=1.2, eps=1e-8, tol=0 (fixed iteration
count).  One "step" = one MM iteration = H-step + W-step + loss (src/nbmf_mm/_solver.py:143-175 of
the reference).  With --gpus N the SAME V is row-sharded over N ranks (strong scaling); every
iteration all-reduces the K x N H-step products over RCCL.

Contract: `python bench.py --gpus N --steps K --warmup W`.  For N>1 it runs one process per GPU: either a
launcher has already started the ranks (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT in the
environment, e.g. by PyTorch's distributed launcher, which is how the driver starts N>1), or -- invoked
plainly -- this script starts N children of itself before anything touches a GPU and waits for them.  The
ranks meet through nbmf_mm_amd._rendezvous (standard library); PyTorch is not imported.  Rank 0 prints ONE
JSON line.
"""
import argparse
import json
import os
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_FP64_MFMA_TFLOPS = 78.6   # MI355X datasheet fp64 matrix = 32 flop/clk/SIMD x 1024 SIMDs x 2.4 GHz
PEAK_HBM_GBPS = 8000.0         # HBM3E, datasheet (MI355X_MICROARCH.md: 8.0 TB/s spec)
ACHIEVABLE_HBM_GBPS = 6300.0   # ... of which a streaming copy reaches 6.29 TB/s (same guide): the line a kernel can be held to
# (the MEASURED peak on the line -- roofline.peak_measured -- is taken on the device that ran the bench, after the timed
#  region: nbmf_selftest_mfma_peak, a ~100 ms loop of nothing but v_mfma_f64_16x16x4_f64 with VGPR accumulators on all SIMDs)


# ---- N > 1: bounds and the retry protocol ---------------------------------------------------------------
# One attempt = one fresh worker process per rank (or per group of ranks).  The process the launcher started stays a
# SUPERVISOR that never touches a GPU: if the attempt fails on ANY rank -- a transport that cannot exchange on this machine,
# a set-up phase that runs into its bound, an exchange that times out in the timed region, a worker that dies -- every
# worker leaves with EXIT_RETRY or an error, and every supervisor starts a second, fresh worker over the host transport,
# whose line says what failed (config.transport_check).  Nothing is ever re-executed inside a process that has used the GPU.
EXIT_RETRY = 75
RDZV_TIMEOUT_S = 60.0          # the ranks of an attempt must all arrive within this
SETUP_BOUND_S = float(os.environ.get("NBMF_BENCH_SETUP_BOUND_S", "100"))   # attach + self-test + trials + verification, summed:
#   enforced by a watchdog thread in every worker (its pieces have bounds of their own -- peer attach: known-answer epochs
#   <= 10 s; each probe wait <= 5 s; verification waits <= 5 s -- except RCCL's communicator set-up, which has none)
ATTEMPT_TIMEOUT_S = float(os.environ.get("NBMF_BENCH_ATTEMPT_TIMEOUT_S", "600"))   # the supervisor's last resort per attempt


class Watchdog:
    """A phase that must end within a bound: `arm(seconds, what, on_fire)` starts the clock, `disarm()` stops it.  When it
    runs out -- the main thread may sit in a device call or a collective for good -- `on_fire(what)` runs on the watchdog's
    thread and must end the process (os._exit)."""

    def __init__(self):
        import threading
        self._cv = threading.Condition()
        self._deadline, self._what, self._on_fire, self.armed_for = None, None, None, 0.0
        threading.Thread(target=self._loop, name="nbmf-bench-watchdog", daemon=True).start()

    def arm(self, seconds, what, on_fire):
        with self._cv:
            self._deadline, self._what, self._on_fire, self.armed_for = time.monotonic() + seconds, what, on_fire, seconds
            self._cv.notify()

    def disarm(self):
        with self._cv:
            self._deadline = None
            self._cv.notify()

    def _loop(self):
        with self._cv:
            while True:
                if self._deadline is None:
                    self._cv.wait()
                    continue
                left = self._deadline - time.monotonic()
                if left > 0:
                    self._cv.wait(left)
                    continue
                fire, what = self._on_fire, self._what
                self._deadline = None
                break
        fire(what)


def write_note(text):
    """Why this worker gives the attempt up: read by its supervisor, handed to the retry, printed on the retry's line."""
    path = os.environ.get("NBMF_BENCH_NOTE")
    if path:
        try:
            with open(path, "w") as f:
                f.write(str(text)[:2000])
        except OSError:
            pass
    print(f"[bench] rank {os.environ.get('RANK', '0')}: {text}", file=sys.stderr, flush=True)


def supervise(argv, worker=None, attempt_timeout=None):
    """The rank process as the launcher started it (N > 1): start a FRESH worker for this rank, wait for it, and if the
    attempt failed start ONE more over the host transport.  Never touches a GPU itself.  `worker`: the command that is
    started (default: this script again; the CPU tests pass a stand-in).  Returns the exit code."""
    import signal
    import tempfile
    worker = list(worker) if worker else [sys.executable, os.path.abspath(__file__)]
    limit = ATTEMPT_TIMEOUT_S if attempt_timeout is None else float(attempt_timeout)
    child = {"p": None}

    def forward(signum, _frame):              # the launcher ends the job: the worker must not outlive its supervisor
        p = child["p"]
        if p is not None and p.poll() is None:
            p.kill()
        raise SystemExit(128 + signum)
    old = {sig: signal.signal(sig, forward) for sig in (signal.SIGTERM, signal.SIGINT)}
    note = tempfile.NamedTemporaryFile(prefix="nbmf_bench_note_", suffix=".txt", delete=False)
    note.close()
    failed = None
    try:
        for attempt in (0, 1):
            env = dict(os.environ, NBMF_BENCH_WORKER="1", NBMF_BENCH_ATTEMPT=str(attempt), NBMF_BENCH_NOTE=note.name,
                       NBMF_RDZV_GENERATION=str(attempt))
            args = list(argv)
            if attempt == 1:
                env["NBMF_BENCH_FAILED"] = failed
                args += ["--transport", "host"]
            open(note.name, "w").close()
            p = child["p"] = subprocess.Popen(worker + args, env=env)
            try:
                code = p.wait(limit)
            except subprocess.TimeoutExpired:
                p.kill()
                p.wait()
                code = -9
                with open(note.name, "w") as f:
                    f.write(f"the worker did not end within the supervisor's last-resort limit of {limit:g} s")
            if code == 0:
                return 0
            why = open(note.name).read().strip() or "no reason recorded"
            failed = f"exit code {code}: {why}"
            if attempt == 1 or "--transport host" in " ".join(argv):
                print(f"[bench] rank {os.environ.get('RANK', '0')}: attempt {attempt} failed ({failed}); giving up", file=sys.stderr, flush=True)
                return code if code > 0 else 1
            print(f"[bench] rank {os.environ.get('RANK', '0')}: attempt 0 failed ({failed}); a fresh worker retries over the host transport",
                  file=sys.stderr, flush=True)
        return 1
    finally:
        for sig, h in old.items():
            signal.signal(sig, h)
        try:
            os.remove(note.name)
        except OSError:
            pass


def make_shard(M, N, r0, r1, seed, density=0.25, observed=0.9, masked=True):
    """Rows [r0, r1) of the synthetic V / mask.  Generated per 4096-row block from
    default_rng([seed, block]) so that any sharding sees the same global matrix."""
    X = np.empty((r1 - r0, N), dtype=np.float64)
    Mk = np.empty((r1 - r0, N), dtype=np.bool_) if masked else None
    BLK = 4096
    for b in range(r0 // BLK, (r1 + BLK - 1) // BLK):
        lo, hi = max(r0, b * BLK), min(r1, (b + 1) * BLK)
        g = np.random.default_rng([seed, b])
        blk = g.random((min(BLK, M - b * BLK), N))
        X[lo - r0:hi - r0] = blk[lo - b * BLK:hi - b * BLK] < density
        if masked:
            blk = g.random((min(BLK, M - b * BLK), N))
            Mk[lo - r0:hi - r0] = blk[lo - b * BLK:hi - b * BLK] < observed
    return X, Mk


def init_factors(M, N, K, seed):
    """Reference init rule (_solver.py:102-136): global RNG, W (M,K) then H (K,N), column-normalise W."""
    rs = np.random.RandomState(seed)          # the numbers of np.random.seed(seed) + global draws; private, so that
    W0 = rs.uniform(0.1, 0.9, (M, K))        # several ranks in one process (one thread each) do not share a stream
    H0 = rs.uniform(0.1, 0.9, (K, N))
    W = W0.T / W0.T.sum(axis=0, keepdims=True)
    return np.ascontiguousarray(W), H0


def cpu_baseline(N, K, seed, masked, projection, budget_rows=2048, iters=3):
    """The CPU oracle (NumPy port of the reference iteration) timed on this host on a bounded
    sample: the first `budget_rows` rows of the same V; cost is linear in M at fixed N, K.
    Returns (seconds per iteration, BLAS threads, loss after the 1 + iters iterations it ran)."""
    from oracle import nbmf_oracle as orc
    try:
        import threadpoolctl
        info = threadpoolctl.threadpool_info()
        threads = max([d.get("num_threads", 1) for d in info] or [1])
    except Exception:
        threads = len(os.sched_getaffinity(0))
    step = orc.mm_step_duchi if projection == "duchi" else orc.mm_step
    X, Mk = make_shard(budget_rows, N, 0, budget_rows, seed, masked=masked)
    mask = Mk.astype(np.float64) if masked else None
    W, H = init_factors(budget_rows, N, K, seed)
    W, H = step(X, W, H, mask, 1.2, 1.2)                   # warm-up
    t0 = time.perf_counter()
    for _ in range(iters):
        W, H = step(X, W, H, mask, 1.2, 1.2)
        loss = orc.mm_loss(X, W, H, mask, 1.2, 1.2)
    dt = (time.perf_counter() - t0) / iters
    return dt, threads, float(loss)


def hip_sample_loss(N, K, seed, masked, projection, device, budget_rows=2048, iters=3):
    """The HIP path on the very sample cpu_baseline() ran (same rows, init and 1 + iters iterations): its final
    loss, for the bench line's `parity` field."""
    from nbmf_mm_amd import _hip
    X, Mk = make_shard(budget_rows, N, 0, budget_rows, seed, masked=masked)
    W, H = init_factors(budget_rows, N, K, seed)
    with _hip.Context(budget_rows, N, K, device=device) as ctx:
        ctx.set_hyper(1.2, 1.2, 1e-8, _hip.PROJ_DUCHI if projection == "duchi" else _hip.PROJ_NORMALIZE)
        ctx.upload(X, mask=Mk)
        ctx.set_factors(W, H)
        losses, _ = ctx.run(1 + iters, 0.0)
    return float(losses[-1])


def host_ram_gb():
    try:
        for line in open("/proc/meminfo"):
            if line.startswith("MemTotal:"):
                return round(int(line.split()[1]) / 1048576.0, 1)
    except OSError:
        pass
    return None


def profiled_traffic(M, N, K, masked, world):
    """HBM bytes per launch of the dominant kernel from the committed rocprofv3 PMC passes of this same
    command (profiles/r*_c3_k64_masked.json <- tools/prof_summary.py; the newest round's file wins):
    (2*FETCH_SIZE + WRITE_SIZE) KiB, the x2 being the gfx950 FETCH_SIZE correction of MI355X_MICROARCH.md
    (HBM section).  Only for the default workload on one GPU; otherwise None."""
    if (M, N, K, masked, world) != (65536, 8192, 64, True, 1):
        return None, None
    import glob
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_c3_k64_masked.json")), reverse=True):
        rec = json.load(open(path))
        if "FETCH_SIZE" in rec and "WRITE_SIZE" in rec:
            return (2.0 * rec["FETCH_SIZE"] + rec["WRITE_SIZE"]) * 1024.0, os.path.relpath(path, ROOT)
    return None, None


def verify_transport(ctx, group, reset, transport, iters=3):
    """The attached transport against the host transport (sums through pinned host memory, formed in rank order):
    `iters` iterations from the same state must give the same losses and the same replicated factor to rounding.
    Setup work, untimed.  A transport that fails is replaced by RCCL (then the host transport), and the line says so:
    the library's own exchange kernels have only ever been run between processes sharing one GPU, so the first run
    over real links checks itself.  Returns (transport in use, verdict string)."""
    from nbmf_mm_amd import _dist

    def curve():
        reset()
        losses, _ = ctx.run(iters, 0.0)
        return np.array(losses), ctx.get_factors()[1]

    try:
        la, Ha = curve()
        ok = True
    except Exception:                       # a timed-out exchange: NBMFHipError
        la, Ha, ok = None, None, False
    ctx.comm_detach()
    _dist.attach_comm(ctx, group, "host")
    lb, Hb = curve()
    ctx.comm_detach()
    if ok:
        ok = bool(np.allclose(la, lb, rtol=1e-12, atol=0) and np.allclose(Ha, Hb, rtol=0, atol=1e-12))
    ok = group.agree(ok)
    if ok:
        _dist.attach_comm(ctx, group, transport)
        reset()
        return transport, "equals the host transport to 1e-12 over %d iterations" % iters
    used = _dist.attach_comm(ctx, group, "rccl" if not transport.startswith("rccl") else "host")
    reset()
    return used, f"{transport} DISAGREED with the host transport and was replaced by {used}"


def launch_ranks(n, per_process=1):
    """`bench.py --gpus N` invoked plainly: start the ranks as children of this script BEFORE this process has
    touched a GPU (it never does), wait for them, and pass on the worst exit code.  One process per rank, or
    (--ranks-per-process R) one process per R consecutive ranks, each rank a host thread with its own context and
    stream: a rehearsal box admits only a few GPU processes at once.  Rank 0 prints the JSON line on the stdout it
    inherits."""
    from nbmf_mm_amd import _rendezvous
    import secrets
    port = _rendezvous.free_port()
    secret = os.environ.get("NBMF_RDZV_SECRET") or secrets.token_hex(16)    # only this job's ranks may join its rendezvous
    procs = []
    for r in range(0, n, per_process):
        here = min(per_process, n - r)
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), NBMF_RDZV_SECRET=secret, NBMF_RANKS_IN_PROCESS=str(here))
        if here > 1:
            # the ranks of one process wait for each other INSIDE kernels: their streams must not share a hardware queue
            env.setdefault("GPU_MAX_HW_QUEUES", str(max(8, 4 * here)))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env))
    rc = 0
    try:
        pending = list(procs)
        while pending:
            for p in list(pending):
                code = p.poll()
                if code is None:
                    continue
                pending.remove(p)
                if code != 0:                         # one rank failed: the others would wait for it forever
                    rc = rc or code
                    for q in pending:
                        q.terminate()
            time.sleep(0.05)
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
    return rc


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--M", type=int, default=65536)
    ap.add_argument("--N", type=int, default=8192)
    ap.add_argument("--K", type=int, default=64)
    ap.add_argument("--event-stride", type=int, default=0,
                    help="time the sweeps of every n-th iteration only (0 = chosen from the warm-up's step time: 1 at >= 3 ms)")
    ap.add_argument("--no-mask", action="store_true")
    ap.add_argument("--projection", default="duchi", choices=["normalize", "duchi"])
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--weak", action="store_true",
                    help="weak scaling: --M rows PER GPU (e.g. --M 32768 --gpus 8 = BASELINE configs[3], 262144 x 8192)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--transport", default="auto", choices=["auto", "peer", "peer2", "rccl", "rccl2", "host"],
                    help="exchange transport for --gpus > 1: peer = the library's own kernels over xGMI (HIP IPC), rccl = RCCL "
                         "all-reduce, host = gloo through pinned memory (tests only); rccl2 = RCCL in two "
                         "overlapped panels; auto = peer and rccl are each timed over five iterations before the run and the faster "
                         "one is kept")
    ap.add_argument("--no-events", action="store_true", help="do not record HIP events around the pass kernels (overhead check)")
    ap.add_argument("--device-data", action="store_true",
                    help="generate the synthetic V / mask on the device (nbmf_generate) instead of uploading host arrays: "
                         "for shapes whose float64 host array is impractical (configs[4]); no CPU baseline")
    ap.add_argument("--overlap", action="store_true",
                    help="row-split exchange in two panels, second one overlapped with compute (sets NBMF_OVERLAP=1)")
    ap.add_argument("--force-comm", action="store_true",
                    help="attach a 1-rank communicator (RCCL, or the peer transport with --transport peer) even with --gpus 1: "
                         "the sharded code path and its per-iteration overhead, minus the wires")
    ap.add_argument("--share-gpu", action="store_true", help="all ranks use device 0 (rehearsal on a 1-GPU box)")
    ap.add_argument("--ranks-per-process", type=int, default=1,
                    help="with the built-in launcher: host this many consecutive ranks in each process, one thread, context and "
                         "stream per rank (the peer transport then addresses same-process arenas directly); default one process "
                         "per rank")
    ap.add_argument("--storage", default="auto", choices=["auto", "f64", "f64w"],
                    help="storage path of V on the device (nbmf_set_storage): auto = 1-byte tile codes for binary data; f64 = doubles, "
                         "the arithmetic the reference applies to real-valued V (two quotients and two logarithms per entry); f64w = "
                         "doubles plus float64 weight tiles (a real-valued mask)")
    ap.add_argument("--no-f64-leg", action="store_true",
                    help="skip the extra leg that times the same data on the 8-byte storage path (f64_storage on the line)")
    ap.add_argument("--no-u8-leg", action="store_true",
                    help="skip the extra leg that uploads the same V as uint8 (upload.uint8 on the line)")
    ap.add_argument("--no-verify", action="store_true",
                    help="skip the untimed check of the chosen transport against the host transport (--gpus > 1)")
    args = ap.parse_args()

    if args.overlap:
        os.environ["NBMF_OVERLAP"] = "1"
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        raise SystemExit(launch_ranks(args.gpus, max(1, args.ranks_per_process)))   # plain invocation: this process only starts the ranks
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    here = int(os.environ.get("NBMF_RANKS_IN_PROCESS", "1"))
    if world > 1 and not os.environ.get("NBMF_BENCH_WORKER"):
        raise SystemExit(supervise(sys.argv[1:]))         # (this process never touches a GPU)
    attempt = int(os.environ.get("NBMF_BENCH_ATTEMPT", "0"))
    import traceback

    def give_up(t=None):
        # the attempt is over for this worker: say why, and leave so that the supervisor can start the retry (attempt 0) or
        # report the failure (the retry itself)
        traceback.print_exc()
        if world > 1 and attempt == 0:
            write_note("".join(traceback.format_exception_only(*sys.exc_info()[:2])).strip())
            os._exit(EXIT_RETRY)
        os._exit(1)
    if here == 1:
        try:
            return run_rank(args, rank, local_rank, world)
        except SystemExit:
            raise
        except BaseException:
            if world == 1:
                raise
            give_up()
    # several ranks in this process: one thread each; any failure ends the process (and with it the attempt)
    import threading
    failed = []

    def body(t):
        try:
            run_rank(args, rank + t, local_rank + t, world)
        except BaseException:
            failed.append(t)
            give_up(t)
    threads = [threading.Thread(target=body, args=(t,)) for t in range(here)]
    for th in threads:
        th.start()
    for th in threads:
        th.join()
    if failed:
        raise SystemExit(1)


def run_rank(args, rank, local_rank, world):
    from nbmf_mm_amd import _dist, _hip, _rendezvous
    group = _rendezvous.init_from_env(timeout=RDZV_TIMEOUT_S, rank=rank, world=world)   # plumbing only: handles / ids, barriers, max-over-ranks
    attempt = int(os.environ.get("NBMF_BENCH_ATTEMPT", "0"))
    dog = Watchdog() if world > 1 else None

    def out_of_time(what):
        # (watchdog thread; the main thread may be stuck for good.)  Attempt 0: the supervisors retry over the host transport.
        write_note(f"set-up exceeded its bound of {dog.armed_for:g} s while: {what}")
        os._exit(EXIT_RETRY if attempt == 0 else 1)

    M, N, K = (args.M * world if args.weak else args.M), args.N, args.K
    masked = not args.no_mask
    r0, r1 = _dist.shard_bounds(M, world, rank)

    W_full, H0 = init_factors(M, N, K, args.seed)
    # one rank = one GPU.  A launcher may have narrowed every rank's view to its own card (then that card is
    # device 0); anything in between -- several cards visible but fewer than ranks -- would silently stack ranks
    # on one GPU, so it is refused.
    n_visible = _hip.device_count()
    if args.share_gpu or n_visible == 1:
        dev_index = 0
    elif local_rank < n_visible:
        dev_index = local_rank
    else:
        raise SystemExit(f"LOCAL_RANK={local_rank} but only {n_visible} GPUs are visible to this rank "
                         f"(need one per rank, or exactly one; --share-gpu stacks all ranks on device 0)")
    ctx = _hip.Context(r1 - r0, N, K, device=dev_index)
    proj = _hip.PROJ_DUCHI if args.projection == "duchi" else _hip.PROJ_NORMALIZE
    ctx.set_hyper(1.2, 1.2, 1e-8, proj)
    t_up = time.perf_counter()
    X = Mk = None
    if args.device_data:
        # every rank generates its own rows of the same global matrix (nbmf_generate_slice)
        ctx.generate(args.seed, density=0.25, observed=0.9 if masked else 1.0, row0=r0, col0=0, n_global=N)
        binary_path, bytes_up = True, 0
        args.no_cpu_baseline = True
    else:
        X, Mk = make_shard(M, N, r0, r1, args.seed, masked=masked)
        t_up = time.perf_counter()                     # the upload alone, not the synthetic generation before it
        ctx.set_storage(args.storage)
        binary_path = ctx.upload(X, mask=(Mk.astype(np.float64) if (masked and args.storage == "f64w") else Mk))
        bytes_up = X.nbytes + (Mk.nbytes if masked else 0)
    t_up = time.perf_counter() - t_up
    f64_leg = (world == 1 and args.storage == "auto" and binary_path and not args.device_data and not args.no_f64_leg
               and not args.force_comm)
    u8_leg = world == 1 and binary_path and not args.device_data and not args.force_comm and not args.no_u8_leg
    if not (f64_leg or u8_leg):
        X = Mk = None

    def reset():
        ctx.set_factors(np.ascontiguousarray(W_full[:, r0:r1]), H0)

    reset()
    transport, trials, transport_check, setup_s, shared, comm_seen = "none", None, None, None, False, None
    if world > 1:
        # everything between here and the timed region that involves a transport -- attach, known-answer self-test, the
        # trials, the verification against the host transport -- is ONE phase with ONE bound (setup_bound_s on the line)
        t_setup = time.perf_counter()
        dog.arm(SETUP_BOUND_S, "selecting and verifying the exchange transport", out_of_time)
        # (fault injection for the tests of this very mechanism: the named rank never leaves the phase in attempt 0)
        if attempt == 0 and os.environ.get("NBMF_BENCH_FAULT") == f"hang_in_setup:{rank}":
            time.sleep(1e6)
        # (physical cards, not indices: a launcher may have narrowed every rank's view to its own card, device 0 everywhere)
        shared = len(set(group.all_gather(_hip.device_bus_id(dev_index)))) < world
        if args.transport == "auto":
            # time a few iterations over each transport that attaches and keep the faster one (setup, untimed)
            # (RCCL refuses two ranks on one device -- and an attempt it refuses leaves its service threads behind, which
            #  on a box with 16 host cores for 8 ranks slowed the timed run sevenfold: where ranks share a card, as in a
            #  rehearsal on one GPU, it is not tried)
            transport, trials = _dist.attach_fastest(ctx, group, reset, candidates=("peer", "peer2") + (() if shared else ("rccl",)))
        else:
            transport = _dist.attach_comm(ctx, group, args.transport)
        transport_check = None
        if transport != "host" and not args.no_verify:
            ctx.set_peer_timeout_ms(5000.0)           # (the verification's waits are probes too; the run gets the default back)
            try:
                transport, transport_check = verify_transport(ctx, group, reset, transport)
            finally:
                ctx.set_peer_timeout_ms(0.0)
        if os.environ.get("NBMF_BENCH_FAILED"):
            # this is the second attempt: say on the line what ended the first one, on which ranks
            notes = group.all_gather(os.environ["NBMF_BENCH_FAILED"])
            transport_check = ("attempt 0 FAILED and this line was measured by fresh workers over the host transport -- " +
                               "; ".join(f"rank {r}: {n}" for r, n in enumerate(notes)))
        info = ctx.comm_info()
        comm_seen = {"kind": info["kind"], "nranks_seen": group.all_gather(info["nranks_seen"])}
        dog.disarm()
        setup_s = group.max_float(time.perf_counter() - t_setup)
    elif args.force_comm:
        transport = _dist.attach_comm(ctx, group, "rccl" if args.transport == "auto" else args.transport) + "(1 rank)"

    def sync():
        ctx.synchronize()                       # the library's own stream
        _hip.device_synchronize(dev_index)      # whole device (hipDeviceSynchronize; contract: barrier + device sync)
        group.barrier()

    def timed(steps):
        sync()
        # (fault injection for the tests of the retry protocol: the named rank fails inside the timed region of attempt 0,
        #  as an exchange that times out there would -- NBMFHipError out of ctx.run)
        if attempt == 0 and world > 1 and os.environ.get("NBMF_BENCH_FAULT") == f"fail_in_run:{rank}":
            raise _hip.NBMFHipError("injected: the exchange timed out in the timed region")
        t0 = time.perf_counter()
        losses, _ = ctx.run(steps, 0.0)
        ctx.synchronize()
        dt = time.perf_counter() - t0
        sync()
        return group.max_float(dt), losses

    event_stride = 1
    losses_first = None          # the losses of the first iterations from the initial factors (the uint8 leg compares with them)
    if args.warmup > 0:
        tw0 = time.perf_counter()
        losses_first, _ = ctx.run(args.warmup, 0.0)
        ctx.synchronize()
        warm_ms = 1e3 * (time.perf_counter() - tw0) / args.warmup
        # a timed dispatch costs ~2.5 us: where an iteration is short, the events ride on a sample of the timed region's
        # sweeps (every 4th iteration below 1.5 ms per step, every 2nd below 3 ms) -- roofline.timed_launches says how many
        if args.event_stride > 0:
            event_stride = args.event_stride
        elif args.steps >= 16:
            event_stride = 4 if warm_ms < 1.5 else (2 if warm_ms < 3.0 else 1)
        event_stride = int(group.max_float(float(event_stride)))
    ctx.timing_enable(False if args.no_events else event_stride)
    dt, losses = timed(args.steps)
    tim = ctx.timing()
    ctx.timing_enable(False)
    h_chunks = ctx.sweep_info()["h_chunks"]
    replicas_identical = None
    devices = group.all_gather(dev_index)
    device_bus_ids = group.all_gather(_hip.device_bus_id(dev_index))
    if world > 1:
        # the replicated factor must be the same bits on every rank whatever the transport did (outside the timed region)
        import hashlib
        digest = hashlib.sha256(ctx.get_factors()[1].tobytes()).hexdigest()
        table = group.all_gather((digest, [float(v) for v in losses]))
        replicas_identical = all(t == table[0] for t in table)
    # the same steps under the other projection: "normalize" is the reference's own code path (_solver.py:54,57),
    # "duchi" the README extension BASELINE configs[2] names -- same kernels, reported side by side
    other = "normalize" if args.projection == "duchi" else "duchi"
    ctx.set_hyper(1.2, 1.2, 1e-8, _hip.PROJ_NORMALIZE if other == "normalize" else _hip.PROJ_DUCHI)
    reset()
    if args.warmup > 0:
        ctx.run(args.warmup, 0.0)
    dt_other, losses_other = timed(args.steps)
    if world == 1:
        ctx.close()               # (N > 1: the context stays for the extra RCCL leg, after the line has been put together)
    # the same data, factors and steps on the 8-byte storage path: what a real-valued V costs (the reference accepts any
    # V in [0, 1], _base.py:90; BASELINE's configs say "fp64 V"): two quotients and two logarithms per entry instead of
    # one reciprocal and no logarithm
    f64 = None
    if f64_leg:
        ctx = _hip.Context(r1 - r0, N, K, device=dev_index)
        ctx.set_hyper(1.2, 1.2, 1e-8, proj)
        ctx.set_storage("f64")
        ctx.upload(X, mask=Mk)
        reset()
        if args.warmup > 0:
            ctx.run(args.warmup, 0.0)
        ctx.timing_enable(not args.no_events)
        dt64, losses64 = timed(args.steps)
        t64 = ctx.timing()
        ctx.close()
        h64 = t64["hpass_ms"] / max(1, t64["hpass_launches"])
        w64 = t64["wpass_ms"] / max(1, t64["wpass_launches"])
        fl = 6.0 * (r1 - r0) * N * K
        f64 = {"value": args.steps / dt64, "unit": "it/s", "storage": "f64 tiles, mask folded in as NaN (8 B per entry and image)",
               "hpass_ms": h64, "wpass_ms": w64,
               "frac": (fl / (h64 * 1e-3) / 1e12 / PEAK_FP64_MFMA_TFLOPS) if h64 > 0 else None,
               "wpass_executed_frac": (4.0 / 6.0 * fl / (w64 * 1e-3) / 1e12 / PEAK_FP64_MFMA_TFLOPS) if w64 > 0 else None,
               "executed_frac": (10.0 / 6.0 * fl * (args.steps / dt64) / 1e12) / PEAK_FP64_MFMA_TFLOPS,
               "hpass_tflops": (fl / (h64 * 1e-3) / 1e12) if h64 > 0 else None,
               "final_nll_per_entry": float(losses64[-1]),
               "rel_nll_vs_u8_path": abs(float(losses64[-1]) - float(losses[-1])) / abs(float(losses[-1]))}

    # the same V handed over as uint8 (one byte per entry, nbmf_upload_v) instead of float64: upload time and rate, and
    # the first iterations of the fit against those of the float64 upload (the same bits)
    up8 = None
    if u8_leg:
        X8 = X.astype(np.uint8)
        X = None
        ctx = _hip.Context(r1 - r0, N, K, device=dev_index)
        ctx.set_hyper(1.2, 1.2, 1e-8, proj)
        t8 = time.perf_counter()
        ctx.upload(X8, mask=Mk)
        t8 = time.perf_counter() - t8
        reset()
        l8, _ = ctx.run(min(3, args.steps + args.warmup), 0.0)
        ctx.close()
        first = (losses_first if losses_first is not None else losses)[:len(l8)]
        nb = X8.nbytes + (Mk.nbytes if masked else 0)
        up8 = {"bytes": nb, "seconds": t8, "GBps_pcie_inclusive": nb / t8 / 1e9,
               "host_array": "uint8 V + bool mask, 1 byte per entry each (nbmf_upload_v)",
               "fit_equals_f64_upload_bitwise": bool(len(first) == len(l8) and all(float(a) == float(b) for a, b in zip(first, l8)))}
        X8 = None
    X = Mk = None
    peak_meas = _hip.mfma_peak(dev_index, 100.0) if rank == 0 else None

    if rank == 0:
        its = args.steps / dt
        m_loc = r1 - r0
        h_ms = tim["hpass_ms"] / max(1, tim["hpass_launches"])
        w_ms = tim["wpass_ms"] / max(1, tim["wpass_launches"])
        # algorithmic flop of one H-pass launch: Theta + two back-products = 6*m*N*K
        flop_pass = 6.0 * m_loc * N * K
        achieved = flop_pass / (h_ms * 1e-3) / 1e12 if h_ms > 0 else 0.0
        traffic, traffic_src = profiled_traffic(M, N, K, masked, world)
        # ... and its algorithmic HBM bytes: the data image the sweep reads (1/4 byte per entry as lane-mask records, 8 as
        # doubles, 16 with weight tiles) + one read of the streamed factor's two operand images and of the stationary one +
        # the slab writes.  Which line binds: the larger of the two lower bounds on the
        # launch's time, flop / 78.6 TFLOP/s and bytes / 6.3 TB/s (the achievable HBM rate).
        bytes_per_entry = 0.25 if binary_path else (16.0 if args.storage == "f64w" else 8.0)
        bytes_pass = m_loc * N * bytes_per_entry + 2.0 * h_chunks * K * N * 8 + 2.0 * m_loc * K * 8 + N * K * 8.0
        t_mfma = flop_pass / (PEAK_FP64_MFMA_TFLOPS * 1e12)
        t_hbm = bytes_pass / (ACHIEVABLE_HBM_GBPS * 1e9)
        bound = "hbm" if t_hbm > t_mfma else "mfma"
        hbm_gbps = bytes_pass / (h_ms * 1e-3) / 1e9 if h_ms > 0 else 0.0
        if f64 and f64.get("hpass_tflops"):
            f64["frac_of_measured"] = f64["hpass_tflops"] / peak_meas["tflops"]
        by_proj = {args.projection: its, other: args.steps / dt_other}
        out = {
            "metric": "MM-iterations/sec", "value": its, "unit": "it/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": 1e3 * dt / args.steps, "higher_is_better": True,
            "scaling": "weak" if args.weak else "strong", "vs_baseline": None, "dtype": "f64", "data": "synthetic (generated on device)" if args.device_data else "synthetic",
            "library": {"abi": _hip.load().nbmf_abi_version(), "source_hash": _hip.source_hash(), "tree_source_hash": _hip.tree_source_hash()},
            "final_nll_per_entry": float(losses[-1]), "replicas_identical": replicas_identical,
            "loss_monotone": bool(all(losses[i] <= losses[i - 1] + 1e-12 for i in range(1, len(losses)))),
            "normalize_value": by_proj["normalize"], "duchi_value": by_proj["duchi"],
            "f64_storage_value": f64["value"] if f64 else None, "f64_storage": f64,
            "final_nll_per_entry_" + other: float(losses_other[-1]),
            "config": {"workload": f"NBMF-MM fit, dense binary V {M}x{N} (float64 API, density 0.25), K={K}, "
                                   f"{'mask 90% observed' if masked else 'no mask'}, projection={args.projection}, "
                                   f"alpha=beta=1.2, tol=0 (BASELINE.json configs[2])",
                       "note": "projection=duchi is the README-only extension BASELINE configs[2] names (no reference code: "
                               "property-tested, parity unpinned); normalize_value is the reference's own path "
                               "(_solver.py:54,57) timed in the same run on the same data",
                       "M": M, "N": N, "K": K, "rows_per_gpu": m_loc, "storage": "lane-mask records for the sweeps (2 bits per entry) + u8 tile codes for the per-lane kernels" if binary_path else ("f64 tiles + f64 weight tiles" if args.storage == "f64w" else "f64 tiles"),
                       "devices": devices, "device_bus_ids": device_bus_ids,
                       "transport": transport, "transport_trials_s_per_5_iterations": trials, "transport_check": transport_check,
                       "transport_sees": comm_seen, "attempt": attempt,
                       "setup_s": setup_s, "setup_bound_s": SETUP_BOUND_S if world > 1 else None,
                       "setup_bound_covers": ("attach + known-answer self-test + trials of every candidate + verification against the host "
                                              "transport, summed; a watchdog in every worker ends an attempt that exceeds it and fresh "
                                              "workers retry over the host transport") if world > 1 else None,
                       "sharding": (f"rows/{world} ({transport}: " + ("reduce-scatter of 2*K*N+1 doubles fused with the H-update, K*N back"
                                                             if transport.startswith("peer") else "all-reduce of 2*K*N+1 doubles"
                                                             + (" in two overlapped panels" if transport.endswith("2") else ""))
                                    + " per iteration)") if world > 1 else "none"},
            "roofline": {"bound": bound, "kernel": "pass_kernel<MODE_H> (fused Theta + ratios + 2 back-products + loglik)",
                         "achieved": achieved if bound == "mfma" else hbm_gbps,
                         "peak": PEAK_FP64_MFMA_TFLOPS if bound == "mfma" else PEAK_HBM_GBPS,
                         "unit": "TFLOP/s" if bound == "mfma" else "GB/s",
                         "frac": achieved / PEAK_FP64_MFMA_TFLOPS if bound == "mfma" else hbm_gbps / PEAK_HBM_GBPS,
                         # both lines, whichever binds: the fraction of the f64 MFMA peak, and of the HBM rate (datasheet
                         # 8 TB/s and the 6.3 TB/s a streaming copy achieves); `binding_frac` = the launch's lower bound
                         # (the larger of flop / 78.6 T and bytes / 6.3 T) over its measured time
                         "mfma": {"achieved_tflops": achieved, "peak": PEAK_FP64_MFMA_TFLOPS, "frac": achieved / PEAK_FP64_MFMA_TFLOPS,
                                  "algorithmic_flop_per_launch": flop_pass},
                         "hbm": {"achieved_gbps": hbm_gbps, "peak": PEAK_HBM_GBPS, "frac": hbm_gbps / PEAK_HBM_GBPS,
                                 "achievable": ACHIEVABLE_HBM_GBPS, "frac_of_achievable": hbm_gbps / ACHIEVABLE_HBM_GBPS,
                                 "algorithmic_bytes_per_launch": bytes_pass},
                         "binding_frac": (max(t_mfma, t_hbm) / (h_ms * 1e-3)) if h_ms > 0 else None,
                         "peak_measured": peak_meas["tflops"], "frac_of_measured": achieved / peak_meas["tflops"],
                         "peak_measured_how": "nbmf_selftest_mfma_peak on this device after the timed region: %.1f ms of bare "
                                              "v_mfma_f64_16x16x4_f64 (VGPR accumulators, two waves per SIMD): %.2f cycles per MFMA "
                                              "at the nominal 2.4 GHz" % (peak_meas["launch_ms"], peak_meas["cycles_per_mfma_at_2p4GHz"]),
                         "traffic": traffic, "traffic_source": traffic_src,
                         "traffic_unit": "bytes per launch (PMC: (2*FETCH_SIZE + WRITE_SIZE) KiB; compulsory: the data image the sweep "
                                         "reads -- lane-mask records, m*N/4 bytes, on the binary path -- + 2*chunks*K*N*8 slab bytes + one "
                                         "read of the streamed factor's two operand images (2*m*K*8) and of the stationary one (N*K*8) = "
                                         "%.3g; what a launch reads beyond that is those images again, once per XCD round of a chunk's "
                                         "workgroups: DESIGN.md 4.1, 5)" % bytes_pass,
                         "hpass_ms": h_ms, "wpass_ms": w_ms, "timed_launches": int(tim["hpass_launches"]), "event_stride": event_stride,
                         # the whole iteration against the same peak, two ways: EXECUTED MFMA flop (the W-pass runs one
                         # back-product instead of two, SURVEY N4: 6 + 4 = 10*m*N*K) -- the utilisation figure -- and the
                         # reference's ALGORITHMIC 12*m*N*K, which credits work that is not executed
                         "executed_frac": (10.0 * m_loc * N * K * its / 1e12) / PEAK_FP64_MFMA_TFLOPS,
                         "wpass_executed_frac": (4.0 * m_loc * N * K / (w_ms * 1e-3) / 1e12) / PEAK_FP64_MFMA_TFLOPS if w_ms > 0 else None,
                         "iteration_frac": (12.0 * m_loc * N * K * its / 1e12) / PEAK_FP64_MFMA_TFLOPS},
            "upload": {"bytes": bytes_up, "seconds": t_up, "GBps_pcie_inclusive": (bytes_up / t_up / 1e9) if bytes_up else None,
                       "host_array": "float64 V + bool mask" if bytes_up else None, "uint8": up8},
        }
        if world == 1 and not args.no_cpu_baseline:
            # the whole matrix needs >= 48 GiB of NumPy temporaries per iteration and ~90 s each; the
            # prescribed sample is an 8192-row block (cost is linear in M at fixed N, K), 1 warm-up + 2 timed iterations
            sample_rows, cpu_iters = min(8192, M), 2
            cdt, threads, closs = cpu_baseline(N, K, args.seed, masked, args.projection, sample_rows, cpu_iters)
            out["cpu_baseline"] = {"value": 1.0 / (cdt * M / sample_rows), "unit": "it/s", "cores": threads,
                                   "kind": "port",
                                   "sample": f"oracle/nbmf_oracle.py (NumPy+OpenBLAS) on the first {sample_rows} rows x {N} cols (SURVEY 8d's "
                                             f"row block), {cpu_iters} iterations after 1 warm-up = {cdt:.2f} s/it, scaled x{M / sample_rows:g} to "
                                             f"{M} rows (cost is linear in M)",
                                   "host_cpus": os.cpu_count(), "affinity": len(os.sched_getaffinity(0)),
                                   "host_ram_gb": host_ram_gb()}
            out["speedup_vs_cpu"] = its / out["cpu_baseline"]["value"]
            # parity number on the line: the HIP path on that same sample against the oracle
            hloss = hip_sample_loss(N, K, args.seed, masked, args.projection, dev_index, sample_rows, cpu_iters)
            out["parity"] = {"rel_nll_vs_oracle": abs(hloss - closs) / abs(closs), "hip_nll": hloss, "oracle_nll": closs,
                             "sample": f"first {sample_rows} rows, same init, {1 + cpu_iters} iterations, projection={args.projection}",
                             "tolerance": 1e-8}
            # ... and on the PINNED path: projection="normalize" is the reference's own code (_solver.py:54,57), the oracle
            # path held bitwise to the reference's golden files (duchi is the README-only extension: parity unpinned)
            if args.projection != "normalize":
                _, _, closs_n = cpu_baseline(N, K, args.seed, masked, "normalize", sample_rows, cpu_iters)
                hloss_n = hip_sample_loss(N, K, args.seed, masked, "normalize", dev_index, sample_rows, cpu_iters)
                out["parity_normalize"] = {"rel_nll_vs_oracle": abs(hloss_n - closs_n) / abs(closs_n), "hip_nll": hloss_n,
                                           "oracle_nll": closs_n, "pinned": "oracle bitwise equal to the reference on tests/golden/*.npz",
                                           "sample": f"first {sample_rows} rows, same init, {1 + cpu_iters} iterations, projection=normalize",
                                           "tolerance": 1e-8}
            else:
                out["parity_normalize"] = dict(out["parity"], pinned="oracle bitwise equal to the reference on tests/golden/*.npz")
    if world > 1:
        # north_star names RCCL; SURVEY 8 f4 names its replacement: the line carries BOTH.  Whatever transport won, RCCL is
        # timed over the same steps from the same state whenever the ranks sit on distinct devices (it refuses two ranks on
        # one), and says itself how many ranks its communicator joins (ncclCommCount).  The line is complete before this leg
        # starts: if the leg runs into its bound the watchdog prints the line without it -- a measurement is never lost to it.
        def add_rccl(rec):
            if rank == 0:
                out["rccl_value"] = rec["value"]
                out["rccl_nranks"] = rec["nranks_seen"]
                out["rccl"] = rec

        def rccl_leg_out_of_time(what):
            write_note(f"the extra RCCL leg exceeded {dog.armed_for:g} s; the line goes out without it")
            if rank == 0:
                add_rccl({"value": None, "nranks_seen": None, "remote": None, "error": f"{what}: no answer within {dog.armed_for:g} s"})
                print(json.dumps(out), flush=True)
            os._exit(0)
        none = {"value": None, "nranks_seen": None, "remote": None}
        if transport in ("rccl", "rccl2"):
            rccl = dict(none, value=args.steps / dt, nranks_seen=comm_seen["nranks_seen"], error=None, note="RCCL is the transport of this line")
        elif attempt > 0 or args.transport == "host":
            rccl = dict(none, error="not timed: this line runs over the host transport")
        elif shared:
            rccl = dict(none, error="not timed: ranks share a device (a rehearsal on one GPU), and RCCL refuses two ranks on one device")
        elif trials is not None and "rccl" not in trials:
            rccl = dict(none, error="RCCL did not attach, or could not exchange, in the selection's trial")
        else:
            dog.arm(SETUP_BOUND_S, "timing RCCL beside the chosen transport", rccl_leg_out_of_time)
            try:
                ctx.comm_detach()
                ctx.set_hyper(1.2, 1.2, 1e-8, proj)
                rccl = _dist.time_transport(ctx, group, reset, "rccl", args.steps, args.warmup)
            except BaseException as e:       # noqa: BLE001  (a rank that died, a broken rendezvous: the line still goes out, and
                dog.disarm()                 #  nobody asks for a retry of an attempt whose measurement is complete)
                add_rccl(dict(none, error=f"{type(e).__name__}: {e}"))
                if rank == 0:
                    print(json.dumps(out), flush=True)
                os._exit(0)
            dog.disarm()
        add_rccl(rccl)
        ctx.close()
    if rank == 0:
        print(json.dumps(out), flush=True)
    try:
        group.close()
    except Exception:                # noqa: BLE001  (the line is out: a rank that is already gone must not turn it into a failure)
        pass


if __name__ == "__main__":
    main()
