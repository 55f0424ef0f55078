#!/bin/bash
# Round 6: what the E[x] and E[log] table writes cost the update kernels (a development build without them: timing only,
# the loss pass and the cold path would read stale tables).
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; O=gpurun_out/r06; mkdir -p $O
for i in 1 2; do for lib in libschpf_hip.so libschpf_hip_dev_notab.so; do
  SCHPF_LIB_PATH=$R/schpf_amd/$lib timeout 300 python tools/explore.py c3 "dtype=f64" "dtype=f32" 2>&1 | grep setting | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); print('$lib', d['setting'], 'iter', d['iter_ms'], 'upd', d['upd_ms'])"
done; done | tee $O/ablate_table_writes_c3.txt
