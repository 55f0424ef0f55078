"""Where the wall-clock of a whole scHPF.fit() at the headline size goes (GPU box): SCHPF_VERBOSE=1 python tools/fitprof.py"""
import sys, time, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, bench
from schpf import scHPF
X = bench.planted_block(100000, 20000, 20, int(100000*20000*0.05*1.6), 42)
import cProfile, pstats
for seed in (1, 2):
    np.random.seed(seed)
    m = scHPF(20, verbose=False)
    t0 = time.perf_counter(); m.fit(X, init="device"); print("fit", time.perf_counter()-t0, len(m.loss))
np.random.seed(2)
m = scHPF(20, verbose=False)
pr = cProfile.Profile(); pr.enable(); m.fit(X, init="device"); pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(22)
