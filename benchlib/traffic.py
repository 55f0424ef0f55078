"""roofline.traffic: HBM-side bytes per sweep launch, measured in the run by rocprofv3 --pmc child passes of bench.py
(one counter per pass, never combined with traces), or attached from the committed record."""
import json
import os
import re
import sys

import numpy as np
from scipy.sparse import coo_matrix

from .setup import init_engine

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")


def _pmc_child(path, K, dtype_name, steps):
    """Body of `bench.py --pmc-child`: the matrix the parent saved, the parent's engine set-up, a few eager
    iterations -- what rocprofv3 counts one PMC counter over.  Prints nothing the parent parses."""
    from schpf_amd import DeviceCAVI
    z = np.load(path)
    X = coo_matrix((z["data"], (z["row"], z["col"])), shape=tuple(int(v) for v in z["shape"]))
    dtype = np.float64 if dtype_name == "f64" else np.float32
    with DeviceCAVI(X.shape[0], X.shape[1], K, dtype=dtype) as eng:
        init_engine(eng, X, K, dtype)
        eng.init_phi_device(12345)
        for _ in range(steps + 2):
            eng.step()
        eng.synchronize()


def live_traffic(X, K, dtype_name, steps=8):
    """HBM-side bytes per sweep launch, MEASURED in this run: two extra child processes of this script under
    `rocprofv3 --pmc` (one counter per pass, never combined with traces; FETCH_SIZE and WRITE_SIZE do not fit
    one pass, MI355X_MICROARCH.md) iterate the same matrix with the same plans; the per-dispatch averages of
    the sweep kernel are read from rocprofv3's rocpd database.  bytes = (2 * FETCH_SIZE + WRITE_SIZE) KiB:
    FETCH_SIZE reports half of the bytes of wide coalesced streams on gfx950 (same guide, HBM section;
    profiles/r01/fetch_size_calibration.txt confirms it for every access shape of this kernel).
    Returns (GB per launch or None, how)."""
    import shutil
    import sqlite3
    import subprocess
    import tempfile
    exe = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(exe):
        return None, "rocprofv3 not found"
    work = tempfile.mkdtemp(prefix="schpf_pmc_", dir="/tmp")
    try:
        path = os.path.join(work, "matrix.npz")
        np.savez(path, data=X.data, row=X.row, col=X.col, shape=np.asarray(X.shape, dtype=np.int64))
        got = {}
        for counter in ("FETCH_SIZE", "WRITE_SIZE"):
            out = os.path.join(work, counter)
            cmd = [exe, "--pmc", counter, "-d", out, "-o", "pmc", "--", sys.executable, BENCH,
                   "--pmc-child", path, "--pmc-k", str(K), "--dtype", dtype_name, "--steps", str(steps)]
            env = dict(os.environ, TMPDIR="/tmp")
            r = subprocess.run(cmd, cwd="/tmp", env=env, capture_output=True, text=True, timeout=300)
            dbs = [os.path.join(d, f) for d, _, fs in os.walk(out) for f in fs if f.endswith(".db")]
            if r.returncode != 0 or not dbs:
                return None, "rocprofv3 --pmc %s failed (rc %d): %s" % (counter, r.returncode, (r.stderr or "")[-200:])
            con = sqlite3.connect(dbs[0])
            rows = con.execute("select kernel_name, count(*), avg(value) from counters_collection where "
                               "counter_name = ? group by kernel_name", (counter,)).fetchall()
            con.close()
            sweep = [(n, c, v) for n, c, v in rows if "tile_sweep_dual_kernel" in n or
                     ("sweep_kernel" in n and "random" not in n and c >= steps)]
            if not sweep:
                return None, "no sweep kernel among the counted dispatches"
            name, count, value = max(sweep, key=lambda t: t[1])
            got[counter] = (name, int(count), float(value))
        gb = (2.0 * got["FETCH_SIZE"][2] + got["WRITE_SIZE"][2]) * 1024.0 / 1e9
        how = ("measured in this run: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (one pass each) over %d launches of %s in a "
               "child process on the same matrix and plans; (2 x %.0f + %.0f) KiB per launch (FETCH_SIZE counts half "
               "of the bytes on gfx950, MI355X_MICROARCH.md)"
               % (got["FETCH_SIZE"][1], re.sub(r"^void |schpf::|\(.*", "", got["FETCH_SIZE"][0]), got["FETCH_SIZE"][2],
                  got["WRITE_SIZE"][2]))
        return gb, how
    except Exception as exc:       # the bench line must not die of its profiler
        return None, "live PMC collection failed: %r" % (exc,)
    finally:
        shutil.rmtree(work, ignore_errors=True)


def static_traffic(config, dtype, info):
    """HBM bytes per sweep launch from the committed rocprofv3 PMC passes (PMC needs its own
    profiler runs, so this is a STATIC figure): attached only when the plan of this run is the
    plan the counters were collected on (same entry slots and partial rows), with the commit."""
    path = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    try:
        with open(path) as fh:
            rec = json.load(fh).get("%s/%s" % (config, dtype))
    except (OSError, ValueError):
        return None, None
    if not rec:
        return None, None
    sig = rec.get("plan", {})
    if any(info.get(k) != v for k, v in sig.items()):
        return None, "profiles/pmc_traffic.json has counters for another plan of %s/%s (stale): not attached" % (config, dtype)
    return rec["bytes_per_launch"] / 1e9, ("static: (2*FETCH_SIZE + WRITE_SIZE) per launch in GB from %s, "
                                           "collected at commit %s on this plan" % (rec.get("source"), rec.get("commit")))
