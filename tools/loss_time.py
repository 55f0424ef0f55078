#!/usr/bin/env python
"""Time the loss evaluation alone (GPU box): python tools/loss_time.py c3 "dtype=f64" "dtype=f64,SCHPF_LOSS_SIDE=0" ...

Settings as in tools/explore.py; prints ms per mean_negative_pois_llh() call (wall, 200 calls after 20 untimed)."""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from schpf_amd import DeviceCAVI  # noqa: E402


def run(X, K, setting):
    kv = dict(item.split("=") for item in setting.split(",") if item)
    dtype = np.float32 if kv.pop("dtype", "f64") == "f32" else np.float64
    for k in list(os.environ):
        if k.startswith("SCHPF_") and k not in ("SCHPF_VERBOSE", "SCHPF_LIB_PATH"):
            del os.environ[k]
    os.environ.update(kv)
    N, G = X.shape
    with DeviceCAVI(N, G, K, dtype=dtype) as eng:
        bench.init_engine(eng, X, K, dtype)
        eng.init_phi_device(1)
        for _ in range(3):
            eng.step()
        for _ in range(20):
            loss = eng.mean_negative_pois_llh()
        eng.synchronize()
        t0 = time.perf_counter()
        for _ in range(200):
            loss = eng.mean_negative_pois_llh()
        ms = (time.perf_counter() - t0) / 200 * 1e3
    print(json.dumps({"setting": setting, "loss_ms": round(ms, 4), "loss": loss}), flush=True)


def main():
    N, G, dens, K = bench.CONFIGS[sys.argv[1]]
    X = bench.synthetic_block(N, G, dens, 42)
    for setting in sys.argv[2:]:
        try:
            run(X, K, setting)
        except Exception as e:
            print(json.dumps({"setting": setting, "error": str(e)}), flush=True)


if __name__ == "__main__":
    main()
