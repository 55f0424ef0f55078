#!/bin/bash
# Round 6: the update kernel's series constants as SCALAR operands (special.h fma_c: one v_fma_f64 per Horner step instead
# of v_mov_b32 x 2 + v_mov_b64 + v_fmac; 114 -> 69 VGPRs, four -> seven waves per SIMD) against the build before
# (schpf_amd/libschpf_hip_base.so): parity subset, then the A/B on one box.
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; O=gpurun_out/r06; mkdir -p $O
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
SCHPF_BACKTRACE=1 timeout 1200 python -m pytest tests/test_ops_gpu.py tests/test_engine_gpu.py -x -q -m gpu -k "ops or iterations_match_oracle or fused_column or random_problems or fit_reproduces or sharded or steps_call or empty_rows or without_stored or largest" > $O/pytest_sconst.log 2>&1; echo "pytest rc $?"; tail -3 $O/pytest_sconst.log | cut -c1-200
for cfg in c3 c4-shard c5-shard c2; do
 for i in 1 2; do for lib in libschpf_hip_base.so libschpf_hip.so; do
  SCHPF_LIB_PATH=$R/schpf_amd/$lib timeout 300 python tools/explore.py $cfg "dtype=f64" "dtype=f32" 2>&1 | grep setting | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); print('$cfg', '$lib', d['setting'], 'iter', d['iter_ms'], 'upd', d['upd_ms'], 'loss', d['loss'])"
 done; done
done | tee $O/ab_scalar_constants.txt
B="--no-cpu-baseline --no-converge --no-traffic --steps 100 --warmup 10"
for i in 1 2; do for lib in libschpf_hip_base.so libschpf_hip.so; do
  SCHPF_LIB_PATH=$R/schpf_amd/$lib python bench.py $B > $O/bench_sconst.json 2>> $O/bench.err
  python - $O/bench_sconst.json $lib <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); r = d["roofline"]
print(sys.argv[2], "value %.1f ms %.4f sweep %.4f upd %.4f sclk %.0f / %.0f" % (d["value"], d["ms_per_step"], r["avg_launch_ms"], r["gamma_updates_ms"], r["sclk_mhz"], r["sclk_mhz_timed_call"]))
PY
done; done | tee -a $O/ab_scalar_constants.txt
