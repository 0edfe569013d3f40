#!/usr/bin/env python3
"""Headline benchmark: MM-iterations/sec of the NBMF-MM inner loop on MI355X.

Workload (BASELINE.json metric / configs[2]): synthetic dense binary V 65536 x 8192 handed over as
float64, K=64, mask with 90 % observed entries, alpha=beta=1.2, eps=1e-8, tol=0 (fixed iteration
count).  One "step" = one MM iteration = H-step + W-step + loss (src/nbmf_mm/_solver.py:143-175 of
the reference).  With --gpus N the SAME V is row-sharded over N ranks (strong scaling); every
iteration all-reduces the K x N H-step products over RCCL.

Contract: `python bench.py --gpus N --steps K --warmup W`; for N>1 it is launched by
torch.distributed.run (one rank per GPU).  Rank 0 prints ONE JSON line.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_FP64_MFMA_TFLOPS = 78.6   # MI355X datasheet fp64 matrix = 32 flop/clk/SIMD x 1024 SIMDs x 2.4 GHz
PEAK_MEASURED_TFLOPS = 71.5        # bare back-to-back v_mfma_f64_16x16x4_f64 on all 1024 SIMDs at the clock the chip holds (tools/microbench.hip)
                               # (measured back-to-back v_mfma_f64_16x16x4_f64: 71.5 TFLOP/s)


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
    np.random.seed(seed)
    W0 = np.random.uniform(0.1, 0.9, (M, K))
    H0 = np.random.uniform(0.1, 0.9, (K, N))
    W = W0.T / W0.T.sum(axis=0, keepdims=True)
    return np.ascontiguousarray(W), H0


def cpu_baseline(N, K, seed, masked, budget_rows=2048, iters=3):
    """The CPU oracle (NumPy port of the reference iteration) timed on this host on a bounded
    sample: the first `budget_rows` rows of the same V; cost is linear in M at fixed N, K."""
    from oracle import nbmf_oracle as orc
    try:
        import threadpoolctl
        info = threadpoolctl.threadpool_info()
        threads = max([d.get("num_threads", 1) for d in info] or [1])
    except Exception:
        threads = len(os.sched_getaffinity(0))
    X, Mk = make_shard(budget_rows, N, 0, budget_rows, seed, masked=masked)
    mask = Mk.astype(np.float64) if masked else None
    W, H = init_factors(budget_rows, N, K, seed)
    W, H = orc.mm_step(X, W, H, mask, 1.2, 1.2)            # warm-up
    t0 = time.perf_counter()
    for _ in range(iters):
        W, H = orc.mm_step(X, W, H, mask, 1.2, 1.2)
        loss = orc.mm_loss(X, W, H, mask, 1.2, 1.2)
    dt = (time.perf_counter() - t0) / iters
    return dt, threads, float(loss)


def profiled_traffic(M, N, K, masked, world):
    """HBM bytes per launch of the dominant kernel from the committed rocprofv3 PMC passes of this same
    command (profiles/r1_c3_k64_masked.json <- tools/prof_summary.py): (2*FETCH_SIZE + WRITE_SIZE) KiB,
    the x2 being the gfx950 FETCH_SIZE correction of MI355X_MICROARCH.md (HBM section).  Only for the
    default workload on one GPU; otherwise None."""
    path = os.path.join(ROOT, "profiles", "r1_c3_k64_masked.json")
    if (M, N, K, masked, world) != (65536, 8192, 64, True, 1) or not os.path.exists(path):
        return None, None
    rec = json.load(open(path))
    if "FETCH_SIZE" not in rec or "WRITE_SIZE" not in rec:
        return None, None
    return (2.0 * rec["FETCH_SIZE"] + rec["WRITE_SIZE"]) * 1024.0, "profiles/r1_c3_k64_masked.json"


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--M", type=int, default=65536)
    ap.add_argument("--N", type=int, default=8192)
    ap.add_argument("--K", type=int, default=64)
    ap.add_argument("--no-mask", action="store_true")
    ap.add_argument("--projection", default="duchi", choices=["normalize", "duchi"])
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--weak", action="store_true",
                    help="weak scaling: --M rows PER GPU (e.g. --M 32768 --gpus 8 = BASELINE configs[3], 262144 x 8192)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--transport", default="auto", choices=["auto", "peer", "rccl", "rccl2", "host"],
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
    args = ap.parse_args()

    if args.overlap:
        os.environ["NBMF_OVERLAP"] = "1"
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus and world > 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if args.gpus > 1 and world == 1:
        raise SystemExit("for --gpus N>1 launch with: python -m torch.distributed.run --nproc-per-node N bench.py --gpus N ...")

    import torch            # plumbing only: rendezvous (gloo), barrier, max-over-ranks; no compute
    import torch.distributed as dist
    from nbmf_mm_amd import _hip

    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("gloo", rank=rank, world_size=world)

    M, N, K = (args.M * world if args.weak else args.M), args.N, args.K
    masked = not args.no_mask
    from nbmf_mm_amd import _dist
    r0, r1 = _dist.shard_bounds(M, world, rank)

    W_full, H0 = init_factors(M, N, K, args.seed)
    # one rank = one GPU; if the launcher has narrowed every rank's view to its own card, that card is device 0
    n_visible = _hip.device_count()
    dev_index = 0 if (args.share_gpu or local_rank >= n_visible) else local_rank
    ctx = _hip.Context(r1 - r0, N, K, device=dev_index)
    ctx.set_hyper(1.2, 1.2, 1e-8, _hip.PROJ_DUCHI if args.projection == "duchi" else _hip.PROJ_NORMALIZE)
    t_up = time.perf_counter()
    if args.device_data:
        if world > 1:
            raise SystemExit("--device-data is a single-GPU measurement option")
        ctx.generate(args.seed, density=0.25, observed=0.9 if masked else 1.0)
        binary_path, bytes_up = True, 0
        args.no_cpu_baseline = True
    else:
        X, Mk = make_shard(M, N, r0, r1, args.seed, masked=masked)
        t_up = time.perf_counter()                     # the upload alone, not the synthetic generation before it
        binary_path = ctx.upload(X, mask=Mk)
        bytes_up = X.nbytes + (Mk.nbytes if masked else 0)
        del X, Mk
    t_up = time.perf_counter() - t_up
    ctx.set_factors(np.ascontiguousarray(W_full[:, r0:r1]), H0)
    transport, trials = "none", None
    if world > 1:
        if args.transport == "auto":
            # time a few iterations over each transport that attaches and keep the faster one (setup, untimed)
            def reset():
                ctx.set_factors(np.ascontiguousarray(W_full[:, r0:r1]), H0)
            transport, trials = _dist.attach_fastest(ctx, dist, reset)
        else:
            transport = _dist.attach_comm(ctx, dist, args.transport)
    elif args.force_comm:
        if args.transport == "peer":
            ctx.comm_init_peer(ctx.peer_export(0), 1, 0)
            transport = "peer(1 rank)"
        else:
            ctx.comm_init(_hip.comm_unique_id(), 1, 0)
            transport = "rccl(1 rank)"

    dev = dev_index
    if torch.cuda.is_available():
        torch.cuda.set_device(dev)

    def sync():
        ctx.synchronize()                       # the library's own stream
        if torch.cuda.is_available():
            torch.cuda.synchronize(dev)         # whole device (contract: barrier + torch.cuda.synchronize())
        if world > 1:
            dist.barrier()

    if args.warmup > 0:
        ctx.run(args.warmup, 0.0)
    ctx.timing_enable(not args.no_events)
    sync()
    t0 = time.perf_counter()
    losses, n_iter = ctx.run(args.steps, 0.0)
    ctx.synchronize()
    dt = time.perf_counter() - t0
    sync()
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    tim = ctx.timing()
    replicas_identical = None
    if world > 1:
        # the replicated factor must be the same bits on every rank whatever the transport did (outside the timed region)
        import hashlib
        digest = hashlib.sha256(ctx.get_factors()[1].tobytes()).hexdigest()
        table = [None] * world
        dist.all_gather_object(table, (digest, [float(v) for v in losses]))
        replicas_identical = all(t == table[0] for t in table)
    ctx.close()

    if rank == 0:
        its = args.steps / dt
        m_loc = r1 - r0
        h_ms = tim["hpass_ms"] / max(1, tim["hpass_launches"])
        w_ms = tim["wpass_ms"] / max(1, tim["wpass_launches"])
        # algorithmic flop of one H-pass launch: Theta + two back-products = 6*m*N*K
        flop_pass = 6.0 * m_loc * N * K
        achieved = flop_pass / (h_ms * 1e-3) / 1e12 if h_ms > 0 else 0.0
        traffic, traffic_src = profiled_traffic(M, N, K, masked, world)
        out = {
            "metric": "MM-iterations/sec", "value": its, "unit": "it/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": 1e3 * dt / args.steps, "higher_is_better": True,
            "scaling": "weak" if args.weak else "strong", "vs_baseline": None, "dtype": "f64", "data": "synthetic (generated on device)" if args.device_data else "synthetic",
            "final_nll_per_entry": float(losses[-1]), "replicas_identical": replicas_identical,
            "loss_monotone": bool(all(losses[i] <= losses[i - 1] + 1e-12 for i in range(1, len(losses)))),
            "config": {"workload": f"NBMF-MM fit, dense binary V {M}x{N} (float64 API, density 0.25), K={K}, "
                                   f"{'mask 90% observed' if masked else 'no mask'}, projection={args.projection}, "
                                   f"alpha=beta=1.2, tol=0 (BASELINE.json configs[2])",
                       "note": "projection=duchi is the README-only extension BASELINE configs[2] names (no reference code: "
                               "property-tested); --projection normalize is the reference path, same kernels and the same "
                               "speed to within 0.2 % (DESIGN.md 5)",
                       "M": M, "N": N, "K": K, "rows_per_gpu": m_loc, "storage": "u8 tile codes" if binary_path else "f64 tiles",
                       "transport_trials_s_per_5_iterations": trials,
                       "sharding": (f"rows/{world} ({transport}: " + ("reduce-scatter of 2*K*N+1 doubles fused with the H-update, K*N back"
                                                             if transport == "peer" else "all-reduce of 2*K*N+1 doubles"
                                                             + (" in two overlapped panels" if transport == "rccl2" else ""))
                                    + " per iteration)") if world > 1 else "none"},
            "roofline": {"bound": "mfma", "kernel": "pass_kernel<MODE_H> (fused Theta + ratios + 2 back-products + loglik)",
                         "achieved": achieved, "peak": PEAK_FP64_MFMA_TFLOPS, "unit": "TFLOP/s",
                         "frac": achieved / PEAK_FP64_MFMA_TFLOPS,
                         "peak_measured": PEAK_MEASURED_TFLOPS, "frac_of_measured": achieved / PEAK_MEASURED_TFLOPS,
                         "traffic": traffic, "traffic_source": traffic_src,
                         "traffic_unit": "bytes per launch (PMC: (2*FETCH_SIZE + WRITE_SIZE) KiB; algorithmic: "
                                         "m*N code bytes + 2*chunks*K*N*8 slab bytes = %.3g)" % (m_loc * N + 2.0 * 16 * K * N * 8),
                         "hpass_ms": h_ms, "wpass_ms": w_ms,
                         "iteration_frac": (12.0 * m_loc * N * K * its / 1e12) / PEAK_FP64_MFMA_TFLOPS},
            "upload": {"seconds": t_up, "GBps_pcie_inclusive": (bytes_up / t_up / 1e9) if bytes_up else None},
        }
        if world == 1 and not args.no_cpu_baseline:
            sample_rows = 2048
            cdt, threads, closs = cpu_baseline(N, K, args.seed, masked, sample_rows, 3)
            out["cpu_baseline"] = {"value": 1.0 / (cdt * M / sample_rows), "unit": "it/s", "cores": threads,
                                   "kind": "port",
                                   "sample": f"oracle/nbmf_oracle.py (NumPy+OpenBLAS) on the first {sample_rows} rows x {N} cols, "
                                             f"3 iterations after 1 warm-up = {cdt:.2f} s/it, scaled x{M // sample_rows} to {M} rows "
                                             f"(cost is linear in M)",
                                   "host_cpus": os.cpu_count(), "affinity": len(os.sched_getaffinity(0))}
            out["speedup_vs_cpu"] = its / out["cpu_baseline"]["value"]
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
