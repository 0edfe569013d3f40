#!/usr/bin/env python3
"""Multi-GPU fit, one process per GPU:

    python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 --master-port 29500 \\
        examples/sharded_fit.py

(any launcher that sets RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT will do: the ranks meet through
nbmf_mm_amd._rendezvous, standard library only -- PyTorch is not imported)

V (here synthetic, 65536 x 8192) is split by rows; every rank holds its block of V and of W, the Beta factor H is
replicated.  The per-iteration sum over ranks travels by the library's own peer kernels over xGMI, by RCCL, or through
host memory ("auto" picks the first that attaches everywhere).  On a one-GPU box add --share-gpu to rehearse with every
rank on device 0.
"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import numpy as np

from nbmf_mm_amd import _dist, _rendezvous


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--M", type=int, default=65536)
    ap.add_argument("--N", type=int, default=8192)
    ap.add_argument("--K", type=int, default=64)
    ap.add_argument("--iters", type=int, default=50)
    ap.add_argument("--share-gpu", action="store_true")
    args = ap.parse_args()
    dist = _rendezvous.init_from_env()
    rank, world = dist.rank, dist.world
    r0, r1 = _dist.shard_bounds(args.M, world, rank)
    g = np.random.default_rng([0, rank])
    V_local = (g.random((r1 - r0, args.N)) < 0.25).astype(np.float64)
    mask_local = g.random((r1 - r0, args.N)) < 0.9
    W_local, H, losses, n_iter = _dist.fit_sharded(
        V_local, (args.M, args.N), r0, args.K, dist, orientation="beta-dir", shard="rows", max_iter=args.iters, tol=0,
        mask_local=mask_local, random_state=0, device=0 if args.share_gpu else int(os.environ.get("LOCAL_RANK", "0")))
    if rank == 0:
        print(f"{world} ranks, {n_iter} iterations: loss {losses[0]:.6f} -> {losses[-1]:.6f}; "
              f"W block {W_local.shape}, H {H.shape}, rows of W sum to {W_local.sum(axis=1).mean():.12f}")
    dist.close()


if __name__ == "__main__":
    main()
