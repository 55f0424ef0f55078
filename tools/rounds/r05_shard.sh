#!/bin/bash
# Round 5: the per-rank iteration of a 1/8 row shard of C3 (one-rank all-reduce) under a few workgroup / task shapes.
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; O=gpurun_out/r05s; mkdir -p $O
B="--config c4-shard --force-sharded --no-cpu-baseline --no-converge --no-traffic"
i=0
for env in "" "SCHPF_WPB=8" "SCHPF_WPB=8 SCHPF_LDS_KB=76" "SCHPF_HALF=0" "SCHPF_TASKS=512" "SCHPF_TASK_ROUNDING=0" "SCHPF_TAPER=0" ""; do
  i=$((i+1))
  env $env python bench.py $B > $O/b$i.json 2>> $O/bench.err
  python - "$O/b$i.json" "$env" <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); r = d["roofline"]
    print("%-32s value %.1f ms %.4f cell %.4f gene %.4f upd %.4f loss %.4f" % (sys.argv[2] or "(default)", d["value"], d["ms_per_step"], r.get("cell_sweep_ms", 0), r.get("gene_sweep_ms", 0), r["gamma_updates_ms"], d["loss_eval_ms"]))
except Exception as e:
    print(sys.argv[2], "unreadable:", e)
PY
done | tee $O/shard_shapes.txt
python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_driver_style_third_box.json 2>> $O/bench.err
python - $O/bench_driver_style_third_box.json <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); r = d["roofline"]
print("driver-style", d["value"], d["ms_per_step"], r["avg_launch_ms"], r["frac"], d["loss_eval_ms"], d["iterations_per_s_with_loss_every_10"], d["convergence"]["fit_wall_s"])
PY
