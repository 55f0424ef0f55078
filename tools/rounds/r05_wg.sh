#!/bin/bash
# Round 5: two 512-thread workgroups per compute unit (76 KiB windows each) against one 1024-thread workgroup.
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; O=gpurun_out/r05; mkdir -p $O
[ -n "$1" ] && export SCHPF_LIB_PATH=$R/schpf_amd/libschpf_hip_dev_$1.so
SCHPF_VERBOSE=1 timeout 900 python tools/explore.py c3 "dtype=f64" "dtype=f64,SCHPF_WPB=8" "dtype=f64,SCHPF_WPB=8,SCHPF_LDS_KB=76" "dtype=f64,SCHPF_WPB=8,SCHPF_LDS_KB=76,SCHPF_BALANCE=1" "dtype=f64" "dtype=f64,SCHPF_WPB=8,SCHPF_LDS_KB=76,SCHPF_BALANCE=1" \
  "dtype=f32" "dtype=f32,SCHPF_WPB=8,SCHPF_LDS_KB=76" "dtype=f32,SCHPF_WPB=8,SCHPF_LDS_KB=76,SCHPF_BALANCE=1" > $O/ab_wg.txt 2> $O/ab_wg.err
grep setting $O/ab_wg.txt | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); print('c3', d.get('setting'), d.get('iter_ms'), d.get('cell_ms'), d.get('gene_ms'), d.get('upd_ms'), d.get('loss'), d.get('error'))"
grep "ELL fill\|task ranges\|balanced windows" $O/ab_wg.err | head -40
