#!/usr/bin/env python3
"""Several GPUs behind the scikit-learn style estimator, ONE process:

    python examples/multi_gpu_estimator.py --gpus 8
    GPU_MAX_HW_QUEUES=16 python examples/multi_gpu_estimator.py --gpus 8 --share-gpu     # rehearsal on a one-GPU box

`NBMF(..., n_gpus=N)` shards the rows of V over N GPUs of this machine: one host thread, context and stream per GPU, the
global NumPy generator seeded and drawn from once (as the single-GPU fit does), one exchange of the K x N H-step products
per iteration over the library's own peer kernels (or RCCL, or host memory), `W_` put back together at the end.  Results
equal the single-GPU fit to 1e-12 (the sums over ranks are formed in another order).  V may be bool / uint8: it then goes
to the devices one byte per entry.  (One process PER GPU: examples/sharded_fit.py.)
"""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import numpy as np

from nbmf_mm_amd import NBMF


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=2)
    ap.add_argument("--M", type=int, default=32768)
    ap.add_argument("--N", type=int, default=4096)
    ap.add_argument("--K", type=int, default=64)
    ap.add_argument("--iters", type=int, default=50)
    ap.add_argument("--share-gpu", action="store_true", help="all ranks on device 0 (needs GPU_MAX_HW_QUEUES >= 2 * gpus)")
    args = ap.parse_args()
    g = np.random.default_rng(0)
    V = g.random((args.M, args.N)) < 0.25                       # bool: one byte per entry up to the device
    mask = g.random((args.M, args.N)) < 0.9
    kw = dict(n_components=args.K, max_iter=args.iters, tol=0, random_state=0)
    t0 = time.perf_counter()
    many = NBMF(n_gpus=args.gpus, devices=[0] * args.gpus if args.share_gpu else None, **kw).fit(V, mask=mask)
    t1 = time.perf_counter()
    one = NBMF(**kw).fit(V, mask=mask)
    t2 = time.perf_counter()
    print(f"{args.gpus} GPUs: {t1 - t0:.2f} s, one GPU: {t2 - t1:.2f} s (upload included); final loss {many.loss_:.12f} vs {one.loss_:.12f}; "
          f"max |W_ difference| {np.abs(many.W_ - one.W_).max():.1e}, max |components_ difference| "
          f"{np.abs(many.components_ - one.components_).max():.1e}")


if __name__ == "__main__":
    main()
