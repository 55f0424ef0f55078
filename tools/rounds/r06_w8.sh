#!/bin/bash
# Round 6: update kernel at eight waves per SIMD (61 VGPRs) against seven (69; libschpf_hip_w7.so) and the build before the
# scalar constants (libschpf_hip_base.so).
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; O=gpurun_out/r06; mkdir -p $O
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
SCHPF_BACKTRACE=1 timeout 1200 python -m pytest tests/test_ops_gpu.py tests/test_engine_gpu.py -x -q -m gpu -k "ops or iterations_match_oracle or fused_column or random_problems or fit_reproduces or steps_call" > $O/pytest_w8.log 2>&1; echo "pytest rc $?"; tail -3 $O/pytest_w8.log | cut -c1-200
for cfg in c3 c4-shard c5-shard c2; do
 for i in 1 2; do for lib in libschpf_hip_base.so libschpf_hip_w7.so libschpf_hip.so; do
  SCHPF_LIB_PATH=$R/schpf_amd/$lib timeout 300 python tools/explore.py $cfg "dtype=f64" "dtype=f32" 2>&1 | grep setting | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); print('$cfg', '$lib', d['setting'], 'iter', d['iter_ms'], 'upd', d['upd_ms'])"
 done; done
done | tee $O/ab_update_waves.txt
B="--no-cpu-baseline --no-converge --no-traffic --steps 100 --warmup 10"
for i in 1 2 3; do for lib in libschpf_hip_base.so libschpf_hip_w7.so libschpf_hip.so; do
  SCHPF_LIB_PATH=$R/schpf_amd/$lib python bench.py $B > $O/bench_w.json 2>> $O/bench.err
  python - $O/bench_w.json $lib <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); r = d["roofline"]
print(sys.argv[2], "value %.1f ms %.4f sweep %.4f upd %.4f sclk %.0f / %.0f" % (d["value"], d["ms_per_step"], r["avg_launch_ms"], r["gamma_updates_ms"], r["sclk_mhz"], r["sclk_mhz_timed_call"]))
PY
done; done | tee -a $O/ab_update_waves.txt
