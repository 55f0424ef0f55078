#!/usr/bin/env python
"""Kernel-tuning harness (GPU box): one synthetic matrix, many plan/kernel settings.

    python tools/explore.py c3 "dtype=f64" "dtype=f64,SCHPF_LPC=1" "dtype=f32,SCHPF_CHUNK=64" ...

Each setting is a comma-separated list; keys starting with SCHPF_ are environment knobs
read by the library at create/upload time, `dtype` selects the model precision.
"""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from schpf_amd import DeviceCAVI  # noqa: E402


def run(X, K, setting, steps=10):
    kv = dict(item.split("=") for item in setting.split(",") if item)
    dtype = np.float32 if kv.pop("dtype", "f64") == "f32" else np.float64
    for k in list(os.environ):
        if k.startswith("SCHPF_") and k != "SCHPF_VERBOSE":
            del os.environ[k]
    for k, v in kv.items():
        os.environ[k] = v
    N, G = X.shape
    t0 = time.perf_counter()
    with DeviceCAVI(N, G, K, dtype=dtype) as eng:
        bench.init_engine(eng, X, K, dtype)
        t_up = time.perf_counter() - t0
        eng.init_phi_device(1)
        for _ in range(2):
            eng.step()
        eng.synchronize()
        eng.profile(True)
        eng.profile_read()
        t1 = time.perf_counter()
        for _ in range(steps):
            eng.step()
        eng.synchronize()
        wall = (time.perf_counter() - t1) / steps * 1e3
        p = eng.profile_read()
        loss = eng.mean_negative_pois_llh()
        info = eng.plan_info()
    b = bench.algorithmic_bytes(X.nnz, N, G, K, np.dtype(dtype).itemsize)
    cell = p["cell_sweep"]["ms"] / steps
    gene = p["gene_sweep"]["ms"] / steps
    upd = p["gamma_updates"]["ms"] / steps
    print(json.dumps({"setting": setting, "iter_ms": round(wall, 4), "cell_ms": round(cell, 4),
                      "gene_ms": round(gene, 4), "upd_ms": round(upd, 4),
                      "frac_hbm": round(b / (wall * 1e-3) / 8e12, 4), "loss": loss,
                      "upload_s": round(t_up, 2),
                      "plan": info}),
          flush=True)


def main():
    cfg = sys.argv[1]
    if cfg == "c3-planted":   # the convergence matrix of bench.py: planted Gamma-Poisson factors, skewed rows and columns
        N, G, K = 100000, 20000, 20
        X = bench.planted_block(N, G, K, int(N * G * 0.05 * 1.6), 42)
    else:
        N, G, dens, K = bench.CONFIGS[cfg]
        X = bench.synthetic_block(N, G, dens, 42)
    print("matrix", X.shape, X.nnz, flush=True)
    for setting in sys.argv[2:]:
        try:
            run(X, K, setting)
        except Exception as e:  # keep going: a bad knob should not lose the rest
            print(json.dumps({"setting": setting, "error": str(e)}), flush=True)


if __name__ == "__main__":
    main()
