"""When do the persistent workgroups of the dual sweep run dry?  (development build with -DSCHPF_ABLATE=9)

    DEVFLAGS=-DSCHPF_ABLATE=9 tools/devbuild.sh a9
    SCHPF_LIB_PATH=schpf_amd/libschpf_hip_dev_a9.so python tools/tail_study.py c3 f64

Prints, per launch, the spread of the workgroups' finishing times relative to the launch start: the share of
CU-time between a workgroup's end and the last one's (what a perfectly balanced schedule would win back)."""
import ctypes
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from schpf_amd import DeviceCAVI, _lib  # noqa: E402

cfg = sys.argv[1] if len(sys.argv) > 1 else "c3"
dtype = np.float32 if (len(sys.argv) > 2 and sys.argv[2] == "f32") else np.float64
N, G, dens, K = bench.CONFIGS[cfg]
X = bench.synthetic_block(N, G, dens, 42)
with DeviceCAVI(N, G, K, dtype=dtype) as eng:
    bench.init_engine(eng, X, K, dtype)
    eng.init_phi_device(1)
    for _ in range(3):
        eng.step()
    n = 256
    buf = np.zeros(n + 1)
    for it in range(4):
        eng.step()
        _lib.check(_lib.load().schpf_debug_read_wave_out(eng._h, buf.ctypes.data_as(ctypes.c_void_p), n + 1))
        end = (buf[:n] - buf[n]) / 100.0          # wall_clock64 ticks at 100 MHz -> microseconds
        end = end[end > 0]
        print("launch %d: %d workgroups; last ends at %.1f us, mean %.1f us, min %.1f us; idle tail %.1f %% of CU-time; "
              "deciles %s" % (it, end.size, end.max(), end.mean(), end.min(), 100 * (1 - end.mean() / end.max()),
                              np.round(np.percentile(end, [10, 30, 50, 70, 90]), 1)))
