#!/bin/bash
# Round 6: the whole -m gpu suite on the tree as it stands, the driver-style line, and the quick bench lines of the other
# configurations (sharded lines with per_rank / ideal_ms).
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; O=gpurun_out/r06${1:+_$1}; mkdir -p $O
python -c "import __graft_entry__ as g; g.build(); g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc $?"; tail -1 $O/smoke.log
timeout 2400 python -m pytest tests/ -x -q -m gpu --durations=15 > $O/pytest_gpu.log 2>&1; echo "pytest rc $?"; tail -25 $O/pytest_gpu.log
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver_style.json 2> $O/bench_driver_style.err; echo "driver-style rc $?"
B="--no-cpu-baseline --no-converge --no-traffic"
python bench.py --dtype f32 --steps 100 --warmup 10 $B > $O/bench_c3_f32.json 2> $O/bench.err
python bench.py --config c2 $B > $O/bench_c2_f64.json 2>> $O/bench.err
python bench.py --config c5-shard --steps 60 --warmup 10 $B > $O/bench_c5shard_f64.json 2>> $O/bench.err
python bench.py --config c5-shard --dtype f32 --steps 60 --warmup 10 $B > $O/bench_c5shard_f32.json 2>> $O/bench.err
for s in c4-shard c4-shard4 c4-shard2; do
  python bench.py --config $s --force-sharded $B > $O/bench_$(echo $s | tr -d '-')_f64_sharded1.json 2>> $O/bench.err
done
for f in $O/bench_*.json; do python - "$f" <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); r = d["roofline"]
    print(sys.argv[1].split("/")[-1], d["dtype"], "value %.1f ms %.4f bound %s sclk %s hbm %.3f valu@sclk %.3f lds %.3f launch_ms %.4f upd %.4f loss_ms %.3f per_rank %s" % (
        d["value"], d["ms_per_step"], r["bound"], r["sclk_mhz"] and round(r["sclk_mhz"]), r["hbm_frac"],
        r.get("fp64_valu_frac_at_sclk", r.get("fp32_valu_frac_at_sclk")), r["lds"]["frac"], r["avg_launch_ms"], r["gamma_updates_ms"],
        d["loss_eval_ms"], d.get("per_rank") and {k: round(v, 4) for k, v in d["per_rank"].items() if isinstance(v, float)}))
except Exception as e:
    print(sys.argv[1], "unreadable:", e)
PY
done
