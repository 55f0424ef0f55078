#!/bin/bash
# Round 6: eight instead of four partial-row loads in flight in the update kernels' fixed-order sums (bit-identical sums).
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; O=gpurun_out/r06; mkdir -p $O
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
for cfg in c3 c5-shard c4-shard c2; do
 for i in 1 2 3; do for lib in libschpf_hip_base.so libschpf_hip.so; do
  SCHPF_LIB_PATH=$R/schpf_amd/$lib timeout 300 python tools/explore.py $cfg "dtype=f64" "dtype=f32" 2>&1 | grep setting | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); print('$cfg', '$lib', d['setting'], 'iter', d['iter_ms'], 'upd', d['upd_ms'], 'loss', d['loss'])"
 done; done
done | tee $O/ab_partial_loads.txt
B="--no-cpu-baseline --no-converge --no-traffic --steps 100 --warmup 10"
for i in 1 2 3; do for lib in libschpf_hip_base.so libschpf_hip.so; do
  SCHPF_LIB_PATH=$R/schpf_amd/$lib python bench.py $B > $O/bench_w.json 2>> $O/bench.err
  python - $O/bench_w.json $lib <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); r = d["roofline"]
print(sys.argv[2], "value %.1f ms %.4f sweep %.4f upd %.4f sclk %.0f / %.0f" % (d["value"], d["ms_per_step"], r["avg_launch_ms"], r["gamma_updates_ms"], r["sclk_mhz"], r["sclk_mhz_timed_call"]))
PY
done; done | tee -a $O/ab_partial_loads.txt
