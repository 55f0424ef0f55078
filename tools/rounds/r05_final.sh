#!/bin/bash
# Round 5, closing measurement on one box: the whole -m gpu suite, the driver-style line, the bench lines of every
# configuration, then kernel stats + traffic + SQ counters for C3 f64 / f32 and the C5 share (f64).
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; O=gpurun_out/r05g; mkdir -p $O
python -c "import __graft_entry__ as g; g.build(); g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc $?"
timeout 2400 python -m pytest tests/ -x -q -m gpu > $O/pytest_gpu.log 2>&1; echo "pytest rc $?"; tail -3 $O/pytest_gpu.log
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver_style.json 2> $O/bench_driver_style.err; echo "driver-style rc $?"
B="--no-cpu-baseline --no-converge --no-traffic"
python bench.py --steps 100 --warmup 10 > $O/bench_c3_f64.json 2> $O/bench.err
python bench.py --dtype f32 --steps 100 --warmup 10 --no-cpu-baseline > $O/bench_c3_f32.json 2>> $O/bench.err
python bench.py --config c2 $B > $O/bench_c2_f64.json 2>> $O/bench.err
python bench.py --config c2 --dtype f32 $B > $O/bench_c2_f32.json 2>> $O/bench.err
python bench.py --config c5-shard --steps 60 --warmup 10 $B > $O/bench_c5shard_f64.json 2>> $O/bench.err
python bench.py --config c5-shard --dtype f32 --steps 60 --warmup 10 $B > $O/bench_c5shard_f32.json 2>> $O/bench.err
python bench.py --config c5 --steps 20 --warmup 3 $B > $O/bench_c5_whole_f64.json 2>> $O/bench.err
python bench.py --force-sharded $B > $O/bench_c3_f64_sharded1.json 2>> $O/bench.err
for s in c4-shard c4-shard4 c4-shard2; do
  python bench.py --config $s --force-sharded $B > $O/bench_$(echo $s | tr -d '-')_f64_sharded1.json 2>> $O/bench.err
done
timeout 600 python bench.py --gpus 2 --same-gpu --backend gloo --comm torch --config c2 $B > $O/bench_two_ranks_one_gpu_gloo_c2.json 2>> $O/bench.err; echo "self-launch rc $?"
for f in $O/bench_*.json; do python - "$f" <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); r = d["roofline"]
    print(sys.argv[1].split("/")[-1], d["dtype"], "value %.1f ms %.4f hbm %.4f valu %s launch_ms %.4f upd %.4f loss_ms %.3f with_loss %s" % (d["value"], d["ms_per_step"], r["frac"], r.get("fp64_valu_frac", r.get("valu", {}).get("frac")), r["avg_launch_ms"], r["gamma_updates_ms"], d["loss_eval_ms"], d.get("iterations_per_s_with_loss_every_10")))
except Exception as e:
    print(sys.argv[1], "unreadable:", e)
PY
done
bash tools/profile_counters.sh r05g c3 f64 20
bash tools/profile_counters.sh r05g c3 f32 20
bash tools/profile_counters.sh r05g c5-shard f64 20
