import os, sys
sys.path.insert(0, "/root/repo")
import numpy as np
from bench import make_shard, init_factors
from nbmf_mm_amd import _hip
M = int(sys.argv[1]); N, K = 8192, 32
X, _ = make_shard(M, N, 0, M, 0, masked=False)
W, H = init_factors(M, N, K, 0)
with _hip.Context(M, N, K) as ctx:
    ctx.set_hyper(1.2, 1.2, 1e-8, 0)
    ctx.upload(X)
    ctx.set_factors(W, H)
    ctx.run(3, 0.0)
    os.environ["NBMF_PASS_TRACE"] = "1"
    ctx.run(2, 0.0)
