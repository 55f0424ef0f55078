#!/bin/bash
# Closing measurement of a round on ONE GPU box (everything a round's profiles/<tag>/ quotes comes from here):
#   tools/profile_round.sh <tag> [full]          -> gpurun_out/<tag>/
# always : the bench lines of every configuration (C3 f64 with CPU baselines, live traffic and the convergence fits; C3 f32,
#          C2, the C5 share, the row shards of C3 through the sharded driver with per_rank / ideal_ms, all of C5), then
#          kernel stats + HBM traffic + SQ counters of C3 f64, C3 f32 and the C5 share (tools/profile_counters.sh: one
#          rocprofv3 pass per counter group, never combined with traces);
# full   : first the build + smoke, the whole `-m gpu` suite and the driver-style line (--steps 20 --warmup 5).
tag=${1:-r00}; mode=${2:-}
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; O=gpurun_out/$tag; mkdir -p $O
if [ "$mode" = "full" ]; then
  python -c "import __graft_entry__ as g; g.build(); g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc $?"; tail -1 $O/smoke.log
  timeout 2400 python -m pytest tests/ -x -q -m gpu --durations=15 > $O/pytest_gpu.log 2>&1; echo "pytest rc $?"; tail -22 $O/pytest_gpu.log | cut -c1-200
  timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver_style.json 2> $O/bench_driver_style.err; echo "driver-style rc $?"
fi
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
    key = "fp64_valu_frac_at_sclk" if d["dtype"] == "f64" else "fp32_valu_frac_at_sclk"
    print(sys.argv[1].split("/")[-1], d["dtype"], "value %.1f ms %.4f bound %s sclk %s hbm %.3f valu@sclk %.3f lds %.3f launch_ms %.4f upd %.4f loss_ms %.3f with_loss %.1f per_rank %s" % (
        d["value"], d["ms_per_step"], r["bound"], r["sclk_mhz"] and round(r["sclk_mhz"]), r["hbm_frac"], r[key], r["lds"]["frac"],
        r["avg_launch_ms"], r["gamma_updates_ms"], d["loss_eval_ms"], d["iterations_per_s_with_loss_every_10"],
        d.get("per_rank") and {k: round(v, 4) for k, v in d["per_rank"].items() if isinstance(v, float)}))
except Exception as e:
    print(sys.argv[1], "unreadable:", e)
PY
done
bash tools/profile_counters.sh $tag c3 f64 20
bash tools/profile_counters.sh $tag c3 f32 20
bash tools/profile_counters.sh $tag c5-shard f64 20
