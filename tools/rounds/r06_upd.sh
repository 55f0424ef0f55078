#!/bin/bash
# Round 6, second GPU call: the update-kernel A/B (SCHPF_UPD: 1 = round-3 kernel, 2 = round-6 kernel at 4 waves per SIMD,
# 3 / 4 = the same forced to 5 / 6 waves per SIMD with spills), the parity tests the new kernel touches, the trajectory
# tests, and bench lines with the three-roof roofline.
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; O=gpurun_out/r06; mkdir -p $O
python -c "import __graft_entry__ as g; g.build(); g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc $?"; tail -2 $O/smoke.log
timeout 900 python -m pytest tests/test_ops_gpu.py tests/test_engine_gpu.py -x -q -m gpu -k "ops or iterations_match_oracle or fused_column or random_problems or fit_reproduces" > $O/pytest_upd.log 2>&1; echo "pytest upd rc $?"; tail -5 $O/pytest_upd.log
for cfg in c3 c5-shard c2 c4-shard; do
  timeout 600 python tools/explore.py $cfg "dtype=f64,SCHPF_UPD=1" "dtype=f64,SCHPF_UPD=2" "dtype=f64,SCHPF_UPD=3" "dtype=f64,SCHPF_UPD=4" "dtype=f32,SCHPF_UPD=1" "dtype=f32,SCHPF_UPD=2" "dtype=f32,SCHPF_UPD=3" "dtype=f64,SCHPF_UPD=1" "dtype=f64,SCHPF_UPD=2" > $O/ab_update_kernel_$cfg.txt 2>&1
  python - $O/ab_update_kernel_$cfg.txt <<'PY'
import json, sys
for l in open(sys.argv[1]):
    try: d = json.loads(l)
    except Exception: continue
    print(sys.argv[1].split("_")[-1], d["setting"], "iter", d["iter_ms"], "upd", d["upd_ms"], "loss", d["loss"])
PY
done
timeout 1500 python -m pytest tests/test_trajectory_gpu.py -q -m gpu --durations=10 > $O/pytest_traj.log 2>&1; echo "pytest traj rc $?"; tail -15 $O/pytest_traj.log
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 --no-converge > $O/bench_driver_style_2.json 2> $O/bench_2.err; echo "bench rc $?"
timeout 600 python bench.py --config c5-shard --steps 30 --warmup 5 --no-converge --no-cpu-baseline > $O/bench_c5shard_f64_2.json 2>> $O/bench_2.err; echo "bench c5 rc $?"
for f in $O/bench_driver_style_2.json $O/bench_c5shard_f64_2.json; do python - $f <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); r = d["roofline"]
print(d["value"], d["ms_per_step"], "bound", r["bound"], "sclk", r["sclk_mhz"], r["sclk_mhz_timed_call"], "hbm", r["hbm_frac"], "valu", r.get("fp64_valu_frac"), r.get("fp64_valu_frac_at_sclk"), "lds", r["lds"]["frac"], r["lds"]["bytes_per_launch"], "upd", r["gamma_updates_ms"], "sweep", r["avg_launch_ms"])
PY
done
