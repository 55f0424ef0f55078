#!/bin/bash
# Quick bench lines on ONE box (no CPU baseline, no convergence fit, no counter passes): tools/bench_quick.sh <tag>
tag=${1:-q}
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; O=gpurun_out
B="--no-cpu-baseline --no-converge --no-traffic"
python bench.py $B > $O/${tag}_bench_c3_f64.json 2> $O/${tag}_bench.err
python bench.py --dtype f32 $B > $O/${tag}_bench_c3_f32.json 2>> $O/${tag}_bench.err
python bench.py --config c5-shard --steps 30 --warmup 5 $B > $O/${tag}_bench_c5shard_f64.json 2>> $O/${tag}_bench.err
python bench.py --config c5-shard --dtype f32 --steps 30 --warmup 5 $B > $O/${tag}_bench_c5shard_f32.json 2>> $O/${tag}_bench.err
python bench.py --config c4-shard --force-sharded $B > $O/${tag}_bench_c4shard_f64_sharded1.json 2>> $O/${tag}_bench.err
python bench.py --config c2 $B > $O/${tag}_bench_c2_f64.json 2>> $O/${tag}_bench.err
for f in $O/${tag}_bench_*.json; do python - "$f" <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); r = d["roofline"]
    print(sys.argv[1].split("/")[-1], d["dtype"], "value %.1f ms %.4f frac %.4f launch_ms %.4f upd %.4f loss_ms %.3f with_loss %s" % (d["value"], d["ms_per_step"], r["frac"], r["avg_launch_ms"], r["gamma_updates_ms"], d["loss_eval_ms"], d.get("iterations_per_s_with_loss_every_10")))
except Exception as e:
    print(sys.argv[1], "unreadable:", e)
PY
done
