#!/bin/bash
# Round 5: what the double-buffered schedule would cost with its copies free (development builds, SCHPF_ABLATE=3:
# no window staging after a task's first; the results are wrong, only the time counts), against the same for the
# shipped schedules -- separates "the copies are not hidden" from "the schedule itself is slower".
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; O=gpurun_out/r05; mkdir -p $O
for lib in plain a3; do
  export SCHPF_LIB_PATH=$R/schpf_amd/libschpf_hip_dev_$lib.so
  echo "== build $lib"
  timeout 600 python tools/explore.py c3 "dtype=f64" "dtype=f64,SCHPF_DB=1" "dtype=f64,SCHPF_BALANCE=1" "dtype=f32" "dtype=f32,SCHPF_DB=1" 2>/dev/null | grep setting | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); print('c3', d.get('setting'), d.get('iter_ms'), d.get('cell_ms'), d.get('error'))"
  timeout 600 python tools/explore.py c5-shard "dtype=f64" "dtype=f64,SCHPF_DB=1" "dtype=f32" "dtype=f32,SCHPF_DB=1" 2>/dev/null | grep setting | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); print('c5-shard', d.get('setting'), d.get('iter_ms'), d.get('cell_ms'), d.get('error'))"
done 2>&1 | tee $O/ab_db_ablate_staging.txt
