#!/bin/bash
# Round 6: the update kernel with two groups per turn (SCHPF_UPD_PAIR=1, loads of both groups in flight before either is
# computed; half as many blocks) against one group per turn (=0): parity subset, then the A/B on one box.
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; O=gpurun_out/r06; mkdir -p $O
python -c "import __graft_entry__ as g; g.build(); g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc $?"; tail -1 $O/smoke.log
SCHPF_UPD_PAIR=1 SCHPF_BACKTRACE=1 timeout 1200 python -m pytest tests/test_ops_gpu.py tests/test_engine_gpu.py -x -q -m gpu -k "ops or iterations_match_oracle or fused_column or random_problems or fit_reproduces or sharded or steps_call or empty_rows" > $O/pytest_pair.log 2>&1; echo "pytest pair rc $?"; tail -4 $O/pytest_pair.log | cut -c1-200
for cfg in c3 c4-shard c5-shard c2; do
  timeout 600 python tools/explore.py $cfg "dtype=f64,SCHPF_UPD_PAIR=0" "dtype=f64,SCHPF_UPD_PAIR=1" "dtype=f32,SCHPF_UPD_PAIR=0" "dtype=f32,SCHPF_UPD_PAIR=1" "dtype=f64,SCHPF_UPD_PAIR=0" "dtype=f64,SCHPF_UPD_PAIR=1" > $O/ab_update_pairs_$cfg.txt 2>&1
  python - $O/ab_update_pairs_$cfg.txt $cfg <<'PY'
import json, sys
for l in open(sys.argv[1]):
    try: d = json.loads(l)
    except Exception: continue
    print(sys.argv[2], d.get("setting"), "iter", d.get("iter_ms"), "upd", d.get("upd_ms"), "loss", d.get("loss"), d.get("error", ""))
PY
done
B="--no-cpu-baseline --no-converge --no-traffic --steps 100 --warmup 10"
for i in 1 2; do for v in 0 1; do
  SCHPF_UPD_PAIR=$v python bench.py $B > $O/bench_pair$v.json 2>> $O/bench.err
  python - $O/bench_pair$v.json $v <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); r = d["roofline"]
print("SCHPF_UPD_PAIR=%s" % sys.argv[2], "value %.1f ms %.4f sweep %.4f upd %.4f sclk %.0f / %.0f" % (d["value"], d["ms_per_step"], r["avg_launch_ms"], r["gamma_updates_ms"], r["sclk_mhz"], r["sclk_mhz_timed_call"]))
PY
done; done | tee $O/ab_update_pairs_bench_c3_f64.txt
