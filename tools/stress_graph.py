#!/usr/bin/env python
"""Stress of the captured-stretch path (GPU box): engines on the C3 matrix created, driven through schpf_steps(1) /
steps(10) / steps(10) with loss evaluations and downloads between them -- the pattern of tests/test_trajectory_gpu.py --
and closed, over and over, with whole scHPF.fit() calls on the C2 matrix in between.  Written to hunt a segmentation
fault the full -m gpu suite hit once in round 6 inside schpf_steps; run with SCHPF_BACKTRACE=1.

    SCHPF_BACKTRACE=1 python tools/stress_graph.py [rounds]
"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from schpf import scHPF  # noqa: E402
from schpf_amd import DeviceCAVI  # noqa: E402


def main():
    rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 20
    X = bench.synthetic_block(100000, 20000, 0.05, 42)
    X2 = bench.synthetic_block(10000, 5000, 0.03, 42)
    t0 = time.time()
    for i in range(rounds):
        dtype = np.float64 if i % 2 == 0 else np.float32
        with DeviceCAVI(X.shape[0], X.shape[1], 20, dtype=dtype) as eng:
            bench.init_engine(eng, X, 20, dtype)
            losses = []
            for n in (1, 10, 10):
                eng.steps(n)
                got = [eng.get_gamma(nm) for nm in ("xi", "theta", "eta", "beta")]
                losses.append(eng.mean_negative_pois_llh())
            assert all(np.isfinite(losses)) and all(np.all(np.isfinite(g[0])) for g in got)
        np.random.seed(i)
        m = scHPF(10, dtype=dtype, verbose=False, max_iter=60)
        m.fit(X2)
        print("round %d %s ok: C3 losses %s, C2 fit %d checks, %.0f s" % (i, np.dtype(dtype).name,
              ["%.6f" % v for v in losses], len(m.loss), time.time() - t0), flush=True)


if __name__ == "__main__":
    main()
