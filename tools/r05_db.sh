#!/bin/bash
# Round 5: double-buffered sub-windows (SCHPF_DB=1) -- parity of the new plan kind, then A/B on one box.
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; O=gpurun_out/r05; mkdir -p $O
timeout 1500 python -m pytest tests/test_engine_gpu.py -x -q -m gpu -k "db or C5share or balanced" > $O/pytest_db.log 2>&1; echo "pytest db rc $?"
tail -15 $O/pytest_db.log
SCHPF_VERBOSE=1 timeout 600 python tools/explore.py c3 "dtype=f64" "dtype=f64,SCHPF_DB=1" "dtype=f64" "dtype=f64,SCHPF_DB=1" "dtype=f64,SCHPF_BALANCE=1" \
   "dtype=f32" "dtype=f32,SCHPF_DB=1" "dtype=f32" "dtype=f32,SCHPF_DB=1" > $O/ab_db_c3.txt 2> $O/ab_db_c3.err; echo "c3 rc $?"
grep -h "setting" $O/ab_db_c3.txt | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); print(d.get('setting'), d.get('iter_ms'), d.get('cell_ms'), d.get('gene_ms'), d.get('upd_ms'), d.get('loss'), d.get('error'))"
SCHPF_VERBOSE=1 timeout 600 python tools/explore.py c5-shard "dtype=f64" "dtype=f64,SCHPF_DB=1" "dtype=f64,SCHPF_SINGLE=0" "dtype=f64,SCHPF_DB=1,SCHPF_SINGLE=0" "dtype=f64" "dtype=f64,SCHPF_DB=1" \
   "dtype=f32" "dtype=f32,SCHPF_DB=1" "dtype=f32" "dtype=f32,SCHPF_DB=1" > $O/ab_db_c5.txt 2> $O/ab_db_c5.err; echo "c5 rc $?"
grep -h "setting" $O/ab_db_c5.txt | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); print(d.get('setting'), d.get('iter_ms'), d.get('cell_ms'), d.get('gene_ms'), d.get('upd_ms'), d.get('loss'), d.get('error'))"
grep -h "ELL fill\|balanced windows\|tile plan" $O/ab_db_c3.err | head -40
grep -h "ELL fill" $O/ab_db_c5.err | head -40
