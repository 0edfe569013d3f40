#!/usr/bin/env python3
"""NBMF_PASS_TRACE=1 on the sweeps of two iterations: per launch one line with the workgroups' prologue / loop / epilogue
times, the spread of their exits and the mean time to finish by wave slot (profiles/r4_c2_attribution.txt).
usage: tools/trace_pass.py M [N K masked(0/1)]     e.g. 8192 (configs[1]);  8192 8192 64 1 (the 8-GPU shard of configs[2])"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from bench import make_shard, init_factors
from nbmf_mm_amd import _hip
M = int(sys.argv[1])
N = int(sys.argv[2]) if len(sys.argv) > 2 else 8192
K = int(sys.argv[3]) if len(sys.argv) > 3 else 32
masked = bool(int(sys.argv[4])) if len(sys.argv) > 4 else False
X, Mk = make_shard(M, N, 0, M, 0, masked=masked)
W, H = init_factors(M, N, K, 0)
with _hip.Context(M, N, K) as ctx:
    ctx.set_hyper(1.2, 1.2, 1e-8, 0)
    ctx.upload(X, mask=Mk)
    ctx.set_factors(W, H)
    ctx.run(3, 0.0)
    os.environ["NBMF_PASS_TRACE"] = "1"
    ctx.run(2, 0.0)
