#!/bin/bash
# Round 6: `bench.py --gpus 2 --config c5` at FULL size with the per-rank draw -- two ranks on the one GPU of a test box
# (gloo carries the exchange: a plumbing run, not a measurement): each rank draws its half of the 5e8-draw matrix, the
# line reports the whole matrix's nnz and what each rank drew.  Then the whole-matrix path (--gpus 1) for the host side.
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; O=gpurun_out/r06; mkdir -p $O
B="--no-cpu-baseline --no-converge --no-traffic"
timeout 1500 python bench.py --gpus 2 --same-gpu --backend gloo --comm torch --config c5 --steps 6 --warmup 2 $B > $O/bench_c5_two_ranks_one_gpu_gloo.json 2> $O/bench_c5_two_ranks.err; echo "two ranks rc $?"
python - $O/bench_c5_two_ranks_one_gpu_gloo.json <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(d["n_gpus"], d["value"], d["ms_per_step"], d["config"]["generation"], d["config"]["workload"][:160], d["loss_after_warmup"], d["loss_after_steps"])
PY
