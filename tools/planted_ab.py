#!/usr/bin/env python
"""GPU box: steady-state iteration time on the PLANTED matrix of bench.py's convergence leg under two
settings of an environment switch, e.g.  python tools/planted_ab.py SCHPF_HALF 0 2"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from schpf_amd import DeviceCAVI  # noqa: E402

var, values = sys.argv[1], sys.argv[2:]
N, G, dens, K = bench.CONFIGS["c3"]
X = bench.planted_block(N, G, K, target_events=int(N * G * dens * 1.6), seed=42)
print("planted", X.shape, X.nnz, flush=True)
for rep in range(2):
    for v in values:
        if v == "default":
            os.environ.pop(var, None)
        else:
            os.environ[var] = v
        for dtype in (np.float64, np.float32):
            with DeviceCAVI(N, G, K, dtype=dtype) as eng:
                bench.init_engine(eng, X, K, dtype)
                eng.init_phi_device(1)
                eng.steps(10)
                eng.synchronize()
                t0 = time.perf_counter()
                eng.steps(50)
                eng.synchronize()
                ms = (time.perf_counter() - t0) / 50 * 1e3
                info = eng.plan_info()
            print(var, v, np.dtype(dtype).name, "ms/iter %.4f" % ms, "ring", info["ring_cell"], info["ring_gene"],
                  "slots", info["entry_slots_cell"], info["entry_slots_gene"], flush=True)
