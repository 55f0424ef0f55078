#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; O=gpurun_out/r05z; mkdir -p $O
B="--no-cpu-baseline --no-converge --no-traffic"
for split in 1 0 1 0; do
SCHPF_LOSS_SPLIT=$split python bench.py $B > $O/b.json 2>> $O/bench.err
python - $split <<'PY'
import json, sys
d = json.loads(open("gpurun_out/r05z/b.json").read().strip().splitlines()[-1]); r = d["roofline"]
print("split", sys.argv[1], d["value"], d["ms_per_step"], r["avg_launch_ms"], d["loss_eval_ms"], d["loss_eval_samples_ms"], d["iterations_per_s_with_loss_every_10"])
PY
done
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver_style.json 2> $O/bench_driver_style.err; echo "driver-style rc $?"
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r05z/bench_driver_style.json").read().strip().splitlines()[-1]); r = d["roofline"]
print(d["value"], d["ms_per_step"], r["bound"], r["frac"], r["valu"]["frac"], d["loss_eval_ms"], d["loss_eval_samples_ms"], d["iterations_per_s_with_loss_every_10"], d["convergence"]["fit_wall_s"], d["cpu_baseline"]["value"])
PY
timeout 600 python -m pytest tests/test_multigpu.py -x -q -m gpu -k "bench" 2>&1 | tail -2
