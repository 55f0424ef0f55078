#!/bin/bash
# Round 5: the loss pass on its own finer tasks + results in pinned host memory; single step counts in the float32 wide-row loop.
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; O=gpurun_out/r05l; mkdir -p $O
timeout 2400 python -m pytest tests/ -x -q -m gpu > $O/pytest_gpu.log 2>&1; echo "pytest rc $?"; tail -5 $O/pytest_gpu.log
B="--no-cpu-baseline --no-converge --no-traffic"
for split in 1 0; do
  export SCHPF_LOSS_SPLIT=$split
  python bench.py $B > $O/bench_c3_f64_split$split.json 2> $O/bench.err
  python bench.py --dtype f32 $B > $O/bench_c3_f32_split$split.json 2>> $O/bench.err
  python bench.py --config c5-shard --steps 60 --warmup 10 $B > $O/bench_c5shard_f64_split$split.json 2>> $O/bench.err
  python bench.py --config c5-shard --dtype f32 --steps 60 --warmup 10 $B > $O/bench_c5shard_f32_split$split.json 2>> $O/bench.err
  python bench.py --config c2 $B > $O/bench_c2_f64_split$split.json 2>> $O/bench.err
  python bench.py --config c4-shard --force-sharded $B > $O/bench_c4shard_f64_sharded1_split$split.json 2>> $O/bench.err
done
unset SCHPF_LOSS_SPLIT
SCHPF_SINGLE=0 python bench.py --config c5-shard --dtype f32 --steps 60 --warmup 10 $B > $O/bench_c5shard_f32_pairs.json 2>> $O/bench.err
for f in $O/bench_*.json; do python - "$f" <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); r = d["roofline"]
    print(sys.argv[1].split("/")[-1], d["dtype"], "value %.1f ms %.4f launch_ms %.4f loss_ms %.4f with_loss %.1f" % (d["value"], d["ms_per_step"], r["avg_launch_ms"], d["loss_eval_ms"], d.get("iterations_per_s_with_loss_every_10")))
except Exception as e:
    print(sys.argv[1], "unreadable:", e)
PY
done
