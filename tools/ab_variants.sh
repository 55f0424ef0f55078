#!/bin/bash
# A/B/C of development builds on ONE box:  tools/ab_variants.sh v0 v1 v2  (schpf_amd/libschpf_hip_dev_<tag>.so)
R=$GRAFT_REPO_ROOT
for i in 1 2; do
  for tag in "$@"; do
    SCHPF_LIB_PATH=$R/schpf_amd/libschpf_hip_dev_$tag.so python $R/tools/explore.py c3 "dtype=f64" "dtype=f32" 2>&1 | grep setting | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); print('$tag', d['setting'], 'sweep', d['cell_ms'], 'iter', d['iter_ms'], 'loss', round(d['loss'], 9))"
  done
done
