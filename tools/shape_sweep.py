#!/usr/bin/env python
"""GPU box: the upload-time heuristics (task ranges from the list-schedule model, half windows) against the
fixed rules on shapes unlike the benchmark's: tall, wide, small and large K.
    python tools/shape_sweep.py"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import bench  # noqa: E402
import explore  # noqa: E402

SHAPES = [  # (cells, genes, density, K)
    (400_000, 5_000, 0.02, 20),
    (20_000, 60_000, 0.03, 20),
    (100_000, 20_000, 0.05, 5),
    (50_000, 20_000, 0.03, 100),
    (200_000, 30_000, 0.005, 20),
]
for N, G, dens, K in SHAPES:
    X = bench.synthetic_block(N, G, dens, 7)
    print("matrix", X.shape, X.nnz, "K", K, flush=True)
    for setting in ("dtype=f64,SCHPF_RANGES=0", "dtype=f64,SCHPF_VERBOSE=1", "dtype=f32,SCHPF_RANGES=0", "dtype=f32,SCHPF_VERBOSE=1"):
        try:
            explore.run(X, K, setting)
        except Exception as e:
            print({"setting": setting, "error": str(e)[:200]}, flush=True)
