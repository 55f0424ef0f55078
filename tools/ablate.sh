#!/bin/bash
# Ablation timings of the step loop on ONE box: tools/ablate.sh a0 a1 ...  (libs built by tools/devbuild.sh)
R=$GRAFT_REPO_ROOT
cfg=${CFG:-c3}
for i in 1 2; do
  for tag in "$@"; do
    SCHPF_LIB_PATH=$R/schpf_amd/libschpf_hip_dev_$tag.so python $R/tools/explore.py $cfg ${SETTINGS:-dtype=f64 dtype=f32} 2>&1 | grep setting | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l)
    print('$tag', d.get('setting'), 'sweep', d.get('cell_ms'), 'gene', d.get('gene_ms'), 'iter', d.get('iter_ms'), 'upd', d.get('upd_ms'), 'loss', d.get('loss'), d.get('error',''))"
  done
done
