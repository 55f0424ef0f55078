#!/bin/bash
# Round 6: hunt for the segmentation fault the full -m gpu suite hit in schpf_steps (C3 trajectory test, after ~590 tests in
# the same process): the suite again with the native backtrace handler on (SCHPF_BACKTRACE=1) and the side-stream column
# sum off; then the A/B of that side stream on the driver-style line.
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; O=gpurun_out/r06; mkdir -p $O
python -c "import __graft_entry__ as g; g.build(); g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc $?"; tail -1 $O/smoke.log
SCHPF_BACKTRACE=1 SCHPF_SIDE_SUM=0 timeout 2400 python -m pytest tests/ -x -q -m gpu --durations=15 > $O/pytest_gpu_bt.log 2>&1; echo "pytest rc $?"
grep -n "schpf_hip\] fatal" -A40 $O/pytest_gpu_bt.log | head -80; tail -30 $O/pytest_gpu_bt.log | cut -c1-200
B="--no-cpu-baseline --no-converge --no-traffic --steps 100 --warmup 10"
for i in 1 2; do for v in 0 1; do
  SCHPF_SIDE_SUM=$v python bench.py $B > $O/bench_side$v.json 2>> $O/bench.err
  python - $O/bench_side$v.json $v <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); r = d["roofline"]
print("SCHPF_SIDE_SUM=%s" % sys.argv[2], "value %.1f ms %.4f sweep %.4f upd %.4f sclk %.0f / %.0f" % (d["value"], d["ms_per_step"], r["avg_launch_ms"], r["gamma_updates_ms"], r["sclk_mhz"], r["sclk_mhz_timed_call"]))
PY
done; done | tee $O/ab_side_stream_colsum.txt
