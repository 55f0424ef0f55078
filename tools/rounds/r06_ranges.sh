#!/bin/bash
# Round 6: the task-range choice again, now that a partial row is cheaper to sum (update kernel at eight waves): gene-side
# ranges around the model's choice, the per-task constant of the model, C3 both dtypes and the C5 share.
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; O=gpurun_out/r06; mkdir -p $O
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
show='
import sys, json
for l in sys.stdin:
    d = json.loads(l); print(sys.argv[1], d["setting"], "iter", d["iter_ms"], "sweep", d["cell_ms"], "upd", d["upd_ms"], "waves", d["plan"]["n_waves_cell"], d["plan"]["n_waves_gene"])'
for i in 1 2; do
 for dt in f64 f32; do
  SCHPF_VERBOSE=1 timeout 900 python tools/explore.py c3 "dtype=$dt" "dtype=$dt,SCHPF_RANGES_GENE=12" "dtype=$dt,SCHPF_RANGES_GENE=14" "dtype=$dt,SCHPF_RANGES_GENE=16" \
    "dtype=$dt,SCHPF_RANGES_GENE=20" "dtype=$dt,SCHPF_RANGES_GENE=22" "dtype=$dt,SCHPF_RANGES_GENE=26" "dtype=$dt,SCHPF_RANGES_CELL=2" \
    "dtype=$dt,SCHPF_TASK_US=2" "dtype=$dt,SCHPF_TASK_US=5" "dtype=$dt,SCHPF_TASK_US=8" "dtype=$dt" 2>$O/ranges_$dt.err | grep setting | python -c "$show" c3
  grep "task ranges" $O/ranges_$dt.err | sort | uniq -c
 done
 timeout 900 python tools/explore.py c5-shard "dtype=f64" "dtype=f64,SCHPF_TASK_US=2" "dtype=f64,SCHPF_TASK_US=5" "dtype=f64,SCHPF_TASK_US=8" "dtype=f64" 2>$O/ranges_c5.err | grep setting | python -c "$show" c5-shard
 grep "task ranges" $O/ranges_c5.err | sort | uniq -c
done | tee $O/ranges_again.txt
