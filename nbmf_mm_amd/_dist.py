"""Row-sharded multi-GPU fit: one process per GPU, V split by contiguous row blocks.

Internal (beta-dir) layout: rank r holds Y[r0:r1, :] and W[:, r0:r1]; H (k x n) is replicated.
Per iteration the only exchange is ONE sum over ranks of the H-step products [P1 | P2 | loglik]
(2*K*N+1 doubles) on the library's own stream: its own peer kernels over xGMI, or RCCL.  The reference has
no distributed counterpart; arithmetic differs from the single-GPU run by summation order only.

The ranks meet through a ``group`` object (``nbmf_mm_amd._rendezvous``: standard library only -- it moves the
128-byte RCCL id or the HIP-IPC handle blocks, agreement flags and a max over ranks; ``init_from_env()`` reads
RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT as a launcher sets them).  Anything with the same small interface
(``world``, ``rank``, ``all_gather``, ``broadcast``, ``barrier``, ``all_reduce``, ``agree``, ``max_float``)
works; the tests wrap PyTorch's gloo backend that way.  Nothing here imports PyTorch.
"""
from __future__ import annotations

import numpy as np

from . import _hip

# a transport that cannot serve this job says so with NBMFHipError or -- for capability limits such as "at most
# 16 ranks" or "too few columns to slice" (NBMF_ERR_ARG) -- ValueError; either way the ranks must still vote
_REFUSED = (_hip.NBMFHipError, ValueError)


def shard_bounds(M: int, world: int, rank: int):
    """Contiguous, balanced row range [r0, r1) of rank `rank` (each context pads its own shard)."""
    if world < 1 or not (0 <= rank < world):
        raise ValueError(f"bad rank {rank} / world {world}")
    if M < world:
        raise ValueError(f"cannot shard {M} rows over {world} ranks")
    return (M * rank) // world, (M * (rank + 1)) // world


def global_init(M, N, K, random_state, W_init=None, H_init=None):
    """The reference's init (src/nbmf_mm/_solver.py:102-136) evaluated identically on every rank:
    seed the global RNG, draw W (M,K) then H (K,N) unless given, column-normalise W.
    Returns W (K,M) and H (K,N)."""
    # (a private legacy generator seeded like np.random.seed(random_state) gives the same numbers as the global
    #  draw and is safe when several ranks live in one process, one thread each; without a seed the global one is used)
    rs = np.random.RandomState(random_state) if random_state is not None else np.random
    if W_init is None:
        W_init = rs.uniform(0.1, 0.9, (M, K))
    if H_init is None:
        H_init = rs.uniform(0.1, 0.9, (K, N))
    W = np.asarray(W_init, dtype=np.float64).T
    W = W / W.sum(axis=0, keepdims=True)
    return np.ascontiguousarray(W), np.ascontiguousarray(H_init, dtype=np.float64)


def attach_comm(ctx, group, transport="auto", shard_axis=0):
    """Join `ctx` to the job described by ``group`` (see the module docstring).

    transport "peer": the library's own exchange kernels over xGMI (HIP-IPC mapped arenas; the H-update is
                      fused into a reduce-scatter).  One process per rank.
    transport "rccl": RCCL all-reduce on the library's stream.
    "peer2" / "rccl2": the same in two column panels, the second one's exchange overlapped with compute (rows
                      split only).
    transport "host": all-reduce through pinned host memory and ``group`` (tests, rehearsal on one GPU).
    transport "auto": peer, else RCCL, else host -- after each attempt the ranks agree (one exchange of a flag
                      over ``group``) whether it worked everywhere, so the job never splits.
    Returns the transport used.
    """
    world, rank = group.world, group.rank

    def host():
        ctx.comm_init_host(lambda arr: group.all_reduce(arr, "sum"), world, rank, shard_axis)
        return "host"

    def peer():
        handle, err = None, None
        try:
            handle = ctx.peer_export(shard_axis)
        except _REFUSED as e:
            err = str(e)
        table = group.all_gather(handle)
        if any(h is None for h in table):
            return None, err or "another rank could not export its arena"
        try:
            ctx.comm_init_peer(b"".join(table), world, rank, shard_axis)
            ok = True
        except _REFUSED as e:
            ok, err = False, str(e)
        if group.agree(ok):
            return "peer", None
        if ok:
            ctx.comm_detach()
        return None, err or "another rank failed to attach"

    def rccl():
        uid, err = None, None
        if rank == 0:
            try:
                uid = _hip.comm_unique_id()
            except _REFUSED as e:                # librccl could not be loaded
                err = str(e)
        uid = group.broadcast(uid, src=0)
        if uid is None:
            return None, f"RCCL unavailable on rank 0: {err}"
        try:
            ctx.comm_init(uid, world, rank, shard_axis)
            ok = True
        except _REFUSED as e:
            ok, err = False, str(e)
        if group.agree(ok):
            return "rccl", None
        if ok:
            ctx.comm_detach()
        return None, err or "another rank failed to attach"

    if transport == "host":
        return host()
    if transport in ("rccl2", "peer2"):
        # the same transport with the exchange cut into two column panels, the second one travelling while the first
        # is applied and the W-pass starts on its columns (rows split only).  A per-context setting, not an environment
        # variable: several ranks may live in one process.
        ctx.set_exchange_panels(2)
        try:
            attach_comm(ctx, group, transport[:-1], shard_axis)
        finally:
            ctx.set_exchange_panels(0)
        return transport
    if transport not in ("peer", "rccl", "auto"):
        raise ValueError(f"unknown transport {transport!r}")
    errors = []
    for name, attempt in (("peer", peer), ("rccl", rccl)):
        if transport in (name, "auto"):
            used, err = attempt()
            if used:
                return used
            errors.append(f"{name}: {err}")
            if transport == name:
                raise _hip.NBMFHipError("; ".join(errors))
    return host()


def attach_fastest(ctx, group, reset, shard_axis=0, candidates=("peer", "rccl"), iters=5, probe_timeout_ms=5000.0):
    """Attach whichever of ``candidates`` runs the iteration fastest on THIS machine: each one that attaches on
    every rank is timed over ``iters`` iterations (max over ranks) and detached again; the winner is attached
    for good (the host transport if none attaches).  ``reset()`` must restore the factors (``ctx.set_factors``)
    -- it is called before every trial and once more at the end.  Returns ``(transport, {name: seconds})``.
    ("peer2" and "rccl2" may be added to ``candidates``: the two-panel forms win once the exchange itself takes
    longer than about 70 us, which only a machine with real links can tell.)
    While probing, the peer transport's waits are bounded by ``probe_timeout_ms`` instead of the 30 s of a run: a
    transport that attaches but cannot exchange on this machine costs one short timeout, and a family ("peer" and
    "peer2" are the same kernels) that has failed once is not tried again -- the worst case of the whole selection is
    about two probe timeouts plus the RCCL set-up, not minutes."""
    import time
    timings, failed_family = {}, set()
    for name in candidates:
        family = name.rstrip("2")
        if family in failed_family:
            continue
        ctx.set_peer_timeout_ms(probe_timeout_ms)
        try:
            attach_comm(ctx, group, name, shard_axis)
        except _REFUSED:
            failed_family.add(family)
            continue                                  # refused on some rank: every rank got the same answer
        finally:
            ctx.set_peer_timeout_ms(0.0)
        # Every rank walks through the SAME collectives whatever happened to it locally -- agree, (barrier,) max, agree --
        # or a rank whose warm-up failed would answer a healthy rank's barrier with its float, and the job would die
        # instead of moving on to the next transport.
        t, ok = float("inf"), True
        try:
            reset()
            ctx.run(2, 0.0)
            ctx.synchronize()
        except _REFUSED:
            ok = False
        ok = group.agree(ok)                          # a warm-up that failed anywhere: nobody times this transport
        if ok:
            group.barrier()
            try:
                t0 = time.perf_counter()
                ctx.run(int(iters), 0.0)
                ctx.synchronize()
                t = time.perf_counter() - t0
            except _REFUSED:
                ok = False
        t = group.max_float(t)
        ok = group.agree(ok)
        ctx.comm_detach()
        if ok:
            timings[name] = t
        else:
            failed_family.add(family)
    if timings:
        best = min(timings, key=timings.get)
        attach_comm(ctx, group, best, shard_axis)
    else:
        best = attach_comm(ctx, group, "host", shard_axis)
    reset()
    return best, timings


def time_transport(ctx, group, reset, name, steps, warmup=2, shard_axis=0):
    """One timed leg over transport ``name``, whatever transport the job itself runs on: attach it on every rank, ask it what
    it sees (``ctx.comm_info()``), run ``warmup`` + ``steps`` iterations from ``reset()``'s state, detach.  Collective and
    symmetric under partial failure like :func:`attach_fastest`.  Returns the same dict on every rank:
    ``{"value": iterations per second (max time over ranks) or None, "nranks_seen": [what each rank's communicator reports],
    "remote": [...], "error": None or the first rank's message}`` -- ``bench.py`` puts RCCL's on its line (``rccl_value``,
    ``rccl_nranks``) even when another transport won the selection."""
    import time
    out = {"value": None, "nranks_seen": None, "remote": None, "error": None}
    try:
        attach_comm(ctx, group, name, shard_axis)
    except _REFUSED as e:                         # refused on some rank: every rank got the same answer from the vote ...
        msgs = group.all_gather(str(e))           # ... but only the rank(s) it failed on know why: the line quotes those
        own = [f"rank {r}: {m}" for r, m in enumerate(msgs) if "another rank" not in m]
        out["error"] = f"{name} did not attach -- " + "; ".join(own or msgs[:1])
        return out
    info, err, t = {"nranks_seen": None, "remote": None}, None, float("inf")
    try:
        info = ctx.comm_info()
        reset()
        if warmup > 0:
            ctx.run(int(warmup), 0.0)
        ctx.synchronize()
    except _REFUSED as e:
        err = f"warm-up over {name} failed: {e}"
    ok = group.agree(err is None)
    if ok:
        group.barrier()
        try:
            t0 = time.perf_counter()
            ctx.run(int(steps), 0.0)
            ctx.synchronize()
            t = time.perf_counter() - t0
        except _REFUSED as e:
            err = f"timed run over {name} failed: {e}"
    t = group.max_float(t)
    ok = group.agree(err is None)
    ctx.comm_detach()
    table = group.all_gather((info.get("nranks_seen"), info.get("remote"), err))
    out["nranks_seen"] = [row[0] for row in table]
    out["remote"] = [row[1] for row in table]
    if ok:
        out["value"] = int(steps) / t
    else:
        out["error"] = next((f"rank {r}: {row[2]}" for r, row in enumerate(table) if row[2]), "failed on another rank")
    return out


def fit_row_sharded(Y_local, M_global, r0, n_components, group, max_iter=500, tol=1e-5, alpha=1.2, beta=1.2,
                    W_init=None, H_init=None, mask_local=None, random_state=None, eps=1e-8,
                    projection="normalize", device=0, transport="auto"):
    """beta-dir fit of the global (M_global x N) matrix whose rows [r0, r0+len(Y_local)) this rank
    holds.  Every rank must call this.  Returns (W_local (m_local,k), H (k,N), losses, n_iter)."""
    from ._solver import _projection_code, upload_any
    if not hasattr(Y_local, "toarray"):
        Y_local = np.asarray(Y_local, dtype=np.float64)
    m_loc, N = Y_local.shape
    K = int(n_components)
    W, H = global_init(M_global, N, K, random_state, W_init, H_init)
    with _hip.Context(m_loc, N, K, device=device) as ctx:
        ctx.set_hyper(alpha, beta, eps, _projection_code(projection))
        upload_any(ctx, Y_local, mask_local)
        ctx.set_factors(np.ascontiguousarray(W[:, r0:r0 + m_loc]), H)
        attach_comm(ctx, group, transport)
        losses, n_iter = ctx.run(int(max_iter), float(tol))
        Wk, Hk = ctx.get_factors()
    return Wk.T, Hk, [float(v) for v in losses], n_iter


def fit_sharded(V_local, global_shape, offset, n_components, group, orientation="beta-dir", shard="rows",
                max_iter=500, tol=1e-5, alpha=1.2, beta=1.2, W_init=None, H_init=None, mask_local=None,
                random_state=None, eps=1e-8, projection="normalize", device=0, transport="auto", progress=None,
                _register=None):
    """Sharded fit in the user's orientation, V split over the ranks by ``shard`` = "rows"
    (``V_local = V[offset:offset+len, :]``) or "cols" (``V_local = V[:, offset:offset+len]``).

    Which internal axis that is (internal Y = V for beta-dir, V.T for dir-beta, _solver.py:113-123):
      beta-dir/rows and dir-beta/cols split the rows of Y   -> exchange in the H-step (shard_axis 0);
      beta-dir/cols and dir-beta/rows split the columns of Y -> exchange in the W-step (shard_axis 1).
    The factor indexed by the split axis comes back as this rank's slice, the other one whole:
    returns ``(W, H, losses, n_iter)`` with W (rows_here, k) and H (k, cols_here).
    Custom inits are GLOBAL arrays; under dir-beta they are swapped only if BOTH are given (:122-123).
    ``progress(first, losses)``: called from inside the run with every ten finished iterations' losses (the reference
    prints inside its loop, :165-166); the losses are the same on every rank, so one rank's callback is enough.
    """
    from ._solver import _dense, _projection_code, upload_any
    if not hasattr(V_local, "toarray"):
        V_local = _dense(V_local)
    M, N = global_shape
    K = int(n_components)
    if orientation not in ("beta-dir", "dir-beta"):
        raise ValueError(f"Unknown orientation: {orientation}")
    if shard not in ("rows", "cols"):
        raise ValueError(f"shard must be 'rows' or 'cols', got {shard!r}")
    if shard == "rows" and V_local.shape[1] != N or shard == "cols" and V_local.shape[0] != M:
        raise ValueError(f"shard {V_local.shape} does not span the unsplit axis of the global {global_shape}")
    transposed = orientation == "dir-beta"
    m_int, n_int = (N, M) if transposed else (M, N)                  # internal Y is m_int x n_int
    if transposed and W_init is not None and H_init is not None:
        W_init, H_init = np.asarray(H_init).T, np.asarray(W_init).T
    Wi, Hi = global_init(m_int, n_int, K, random_state, W_init, H_init)   # Wi (K, m_int) simplex, Hi (K, n_int) Beta
    split_rows_of_Y = (shard == "rows") != transposed                # axis 0 of the internal matrix
    length = V_local.shape[0] if shard == "rows" else V_local.shape[1]
    sl = slice(offset, offset + length)
    if split_rows_of_Y:
        ctx_shape, W0, H0, axis = (length, n_int), Wi[:, sl], Hi, 0
    else:
        ctx_shape, W0, H0, axis = (m_int, length), Wi, Hi[:, sl], 1
    with _hip.Context(ctx_shape[0], ctx_shape[1], K, device=device) as ctx:
        if _register is not None:
            _register(ctx)                 # (fit_in_process: the thread that joins the ranks may have to cancel this context)
        ctx.set_hyper(alpha, beta, eps, _projection_code(projection))
        upload_any(ctx, V_local, mask_local, transposed=transposed)    # the pack applies the transpose
        ctx.set_factors(np.ascontiguousarray(W0), np.ascontiguousarray(H0))
        attach_comm(ctx, group, transport, shard_axis=axis)
        if progress is not None:
            ctx.set_progress(progress, every=10)
        losses, n_iter = ctx.run(int(max_iter), float(tol))
        Wk, Hk = ctx.get_factors()
    W_out, H_out = (Hk.T, Wk) if transposed else (Wk.T, Hk)           # un-transpose, _solver.py:178-184
    return W_out, H_out, [float(v) for v in losses], n_iter


def fit_in_process(V, n_components, n_gpus, devices=None, orientation="beta-dir", max_iter=500, tol=1e-5, alpha=1.2,
                   beta=1.2, W_init=None, H_init=None, mask=None, random_state=None, verbose=0, eps=1e-8,
                   projection="normalize", transport="auto", _rank_fit=None):
    """``nbmf_mm_solver(..., n_gpus=N)``: the fit of the WHOLE matrix V sharded by rows over ``n_gpus`` GPUs from ONE
    process -- one host thread, context and stream per GPU (the library calls release the GIL), rank r on
    ``devices[r]`` (default: devices 0 .. N-1) with rows ``shard_bounds(M, N, r)`` of V and of the mask (views, no
    copies).  The ranks' exchange arenas are addressed directly (same process: no IPC handles; peer access between the
    devices), RCCL or the host transport otherwise (``transport="auto"``: the first that attaches on every rank).

    Same signature logic and return value as the single-GPU solver (src/nbmf_mm/_solver.py:61-216): the global RNG is
    seeded and drawn from ONCE, in the reference's order (:102-129), the loop runs to ``max_iter`` or the stop rule
    (all ranks decide on the same summed loss), and ``(W (M,k), H (k,N), losses, 0.0, n_iter)`` comes back with the
    split factor's slices put together.  Differs from ``n_gpus=1`` by the order of the sums over ranks only (<= 1e-12).
    Several ranks may name the same device (a rehearsal on one GPU): their kernels wait for each other, so the process
    must have been started with GPU_MAX_HW_QUEUES >= 2 * n_gpus (checked)."""
    import os
    import threading
    from . import _rendezvous
    from ._solver import _draw_init, _projection_code, _touch_up
    _projection_code(projection)
    n_gpus = int(n_gpus)
    if n_gpus < 1:
        raise ValueError("n_gpus must be >= 1")
    if int(max_iter) < 1:
        raise ValueError("max_iter must be >= 1")
    devices = list(range(n_gpus)) if devices is None else [int(d) for d in devices]
    if len(devices) != n_gpus:
        raise ValueError(f"devices names {len(devices)} GPUs, n_gpus is {n_gpus}")
    if _rank_fit is None:
        n_dev = _hip.device_count()
        if any(d < 0 or d >= n_dev for d in devices):
            raise ValueError(f"devices {devices} but {n_dev} GPUs are visible")
        if len(set(devices)) < n_gpus and _hip.hw_queues_at_load() < 2 * n_gpus:
            raise ValueError(f"{n_gpus} ranks on {len(set(devices))} device(s): ranks that share a GPU wait for each other inside "
                             f"kernels, so their streams need hardware queues of their own -- start the process with "
                             f"GPU_MAX_HW_QUEUES={2 * n_gpus} or more (it is read when the HIP runtime starts: "
                             f"{_hip.hw_queues_at_load()} then; setting it now changes nothing)")
    if not hasattr(V, "toarray"):
        from ._solver import _dense
        V = _dense(V)
    elif not hasattr(V, "indptr") or V.format != "csr":
        V = V.tocsr()
    if mask is not None and hasattr(mask, "toarray"):
        mask = mask.tocsr()
    M, N = V.shape
    K = int(n_components)
    shard_bounds(M, n_gpus, 0)                       # (fewer rows than ranks: a ValueError here, not in the rank threads)
    transposed = orientation == "dir-beta"
    if orientation not in ("beta-dir", "dir-beta"):
        raise ValueError(f"Unknown orientation: {orientation}")
    # the reference's seeding and draws, once (:102-129; internal shapes, :113-123), handed to the ranks as GLOBAL inits
    if random_state is not None:
        np.random.seed(random_state)
    m_int, n_int = (N, M) if transposed else (M, N)
    if transposed and W_init is not None and H_init is not None:
        W_init, H_init = np.asarray(H_init).T, np.asarray(W_init).T
    Wi, Hi = _draw_init(m_int, K, n_int, W_init, H_init)                 # (m_int, k), (k, n_int)
    Wi, Hi = np.asarray(Wi, dtype=np.float64), np.asarray(Hi, dtype=np.float64)
    if Wi.shape != (m_int, K) or Hi.shape != (K, n_int):
        raise ValueError(f"operands could not be broadcast together: W_init/H_init give {Wi.T.shape}, {Hi.shape}; "
                         f"expected ({K},{m_int}), ({K},{n_int})")
    W_user, H_user = (Hi.T, Wi.T) if transposed else (Wi, Hi)            # what fit_sharded swaps back (:122-123)
    rank_fit = fit_sharded if _rank_fit is None else _rank_fit
    groups = _rendezvous.LocalGroup.make(n_gpus)
    results, errors = [None] * n_gpus, [None] * n_gpus

    def _report(first, values):
        # the reference prints inside its loop (:165-166): rank 0's context reports the (summed, hence global) losses every
        # ten iterations WHILE the ranks run, as the single-GPU solver does (nbmf_set_progress)
        for it, loss in enumerate(values, start=first):
            if it % 10 == 0:
                print(f"Iter {it:4d}: Loss = {loss:.6f}", flush=True)

    contexts, interrupted = [None] * n_gpus, threading.Event()

    def body(r):
        live = {"progress": _report} if (verbose > 0 and r == 0) else {}
        if _rank_fit is None:
            def register(ctx, r=r):
                contexts[r] = ctx
                if interrupted.is_set():      # the caller was interrupted while this rank was still setting up
                    ctx.cancel()
            live["_register"] = register
        try:
            with groups[r] as g:                                          # (whatever fails in here aborts the group: nobody waits for this rank)
                r0, r1 = shard_bounds(M, n_gpus, r)
                results[r] = rank_fit(V[r0:r1], (M, N), r0, K, g, orientation=orientation, shard="rows", max_iter=max_iter,
                                      tol=tol, alpha=alpha, beta=beta, W_init=W_user, H_init=H_user,
                                      mask_local=None if mask is None else mask[r0:r1], random_state=None, eps=eps,
                                      projection=projection, device=devices[r], transport=transport, **live)
        except BaseException as e:                                       # noqa: B902 (re-raised in the caller's thread)
            errors[r] = e

    threads = [threading.Thread(target=body, args=(r,), name=f"nbmf-rank-{r}", daemon=True) for r in range(n_gpus)]
    for t in threads:
        t.start()
    try:
        for t in threads:
            while t.is_alive():
                t.join(0.2)                  # (a bounded wait: KeyboardInterrupt is delivered between the waits)
    except BaseException:
        # Ctrl-C (or anything else) in the caller: the ranks must not go on sweeping and exchanging behind its back.  Their
        # group is aborted (whoever sits in a collective leaves it), their contexts are cancelled (the runs end at the next
        # iteration; what is enqueued is cut short on the device, the exchange kernels included), and they are given a
        # bounded time to unwind -- closing their contexts, which frees the device memory and the hardware queues -- before
        # the interrupt goes on to the caller.  daemon=True is only the last resort for a rank stuck inside a device call.
        import time
        interrupted.set()
        for g in groups:
            g.abort()
        for ctx in contexts:
            if ctx is not None:
                try:
                    ctx.cancel()
                except Exception:            # noqa: BLE001  (a context that is already closed)
                    pass
        deadline = time.monotonic() + float(os.environ.get("NBMF_INTERRUPT_JOIN_S", "30"))
        for t in threads:
            t.join(max(0.0, deadline - time.monotonic()))
        raise
    # which error to show: a rank's own failure first; among the ConnectionErrors that followed from it (or from a rank
    # that never arrived) the one that NAMES the missing ranks before the generic "another rank has failed"
    first = ([e for e in errors if e is not None and not isinstance(e, ConnectionError)]
             or [e for e in errors if e is not None and "did not reach" in str(e)]
             or [e for e in errors if e is not None])
    if first:
        raise first[0]
    losses, n_iter = results[0][2], results[0][3]
    # V is split by rows in either orientation: every rank comes back with ITS rows of W (the simplex factor under
    # beta-dir, the Beta factor under dir-beta) and with the whole of H, the same bits on every rank
    W = np.concatenate([res[0] for res in results], axis=0)
    H = results[0][1]
    W, H = _touch_up(W, H, orientation)
    losses = [float(v) for v in losses]
    if verbose > 0 and (n_iter < int(max_iter) or (n_iter > 1 and losses[-2] != 0 and
                                                   abs(losses[-2] - losses[-1]) / abs(losses[-2]) < tol)):
        print(f"Converged at iteration {n_iter - 1}")                    # :172-173 (the same rule as the single-GPU solver)
    return W, H, losses, 0.0, n_iter


def fit_restarts(V, n_components, group, n_init, random_state=0, device=0, **solver_kwargs):
    """``n_init`` independent restarts spread over the ranks (replicas: every rank holds all of V and
    runs restarts rank, rank+world, ... with seeds random_state + i); the best final loss wins and its
    factors are broadcast.  Returns ``(W, H, losses, n_iter, best_index)`` on every rank."""
    from ._solver import nbmf_mm_solver
    world, rank = group.world, group.rank
    best = None
    for i in range(rank, int(n_init), world):
        res = nbmf_mm_solver(V, n_components, random_state=random_state + i, device=device, **solver_kwargs)
        if best is None or res[2][-1] < best[1][2][-1]:
            best = (i, res)
    mine = (float("inf"), -1) if best is None else (float(best[1][2][-1]), best[0])
    table = group.all_gather(mine)
    win_rank = min(range(world), key=lambda r: (table[r][0], table[r][1]))
    payload = None
    if rank == win_rank:
        _, (W, H, losses, _, n_iter) = best
        payload = (W, H, losses, n_iter, best[0])
    return group.broadcast(payload, src=win_rank)
