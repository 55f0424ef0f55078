#!/bin/bash
# Round 6: issue priority raised per PHASE of a step (development builds, -DSCHPF_EXP_PRIO=1: s_setprio 1 from the
# reciprocals to the last row request of a step; =2: the other phase) -- static per-wave priorities were within noise in round 4.
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; O=gpurun_out/r06; mkdir -p $O
B="--no-cpu-baseline --no-converge --no-traffic --steps 100 --warmup 10"
for i in 1 2 3; do for v in p0 p1 p2; do for dt in f64 f32; do
  SCHPF_LIB_PATH=$R/schpf_amd/libschpf_hip_dev_$v.so python bench.py $B --dtype $dt > $O/bench_w.json 2>> $O/bench.err
  python - $O/bench_w.json $v <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); r = d["roofline"]
print(sys.argv[2], d["dtype"], "value %.1f ms %.4f sweep %.4f upd %.4f sclk %.0f / %.0f loss %.12f" % (d["value"], d["ms_per_step"], r["avg_launch_ms"], r["gamma_updates_ms"], r["sclk_mhz"], r["sclk_mhz_timed_call"], d["loss_after_steps"]))
PY
done; done; done | tee $O/ab_phase_priority.txt
