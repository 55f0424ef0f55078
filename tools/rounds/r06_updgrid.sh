#!/bin/bash
# Round 6: how many blocks the update launches get (development build with SCHPF_UPD_BLOCKS / SCHPF_UPD_EVEN): 2048 resident
# blocks share 8334 (theta) / 1667 (beta) row groups at C3 -- 4.07 groups per block, i.e. 142 blocks run a fifth round.
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; O=gpurun_out/r06; mkdir -p $O
export SCHPF_LIB_PATH=$R/schpf_amd/libschpf_hip_dev_updgrid.so
show='
import sys, json
for l in sys.stdin:
    d = json.loads(l); print(sys.argv[1], d["setting"], "iter", d["iter_ms"], "upd", d["upd_ms"], "loss", d["loss"])'
for i in 1 2; do
 for cfg in c3 c5-shard; do
  timeout 900 python tools/explore.py $cfg "dtype=f64" "dtype=f64,SCHPF_UPD_EVEN=1" "dtype=f64,SCHPF_UPD_BLOCKS=1024" "dtype=f64,SCHPF_UPD_BLOCKS=1389" \
    "dtype=f64,SCHPF_UPD_BLOCKS=1536" "dtype=f64,SCHPF_UPD_BLOCKS=2084" "dtype=f64,SCHPF_UPD_BLOCKS=2778" "dtype=f64,SCHPF_UPD_BLOCKS=4167" \
    "dtype=f64,SCHPF_UPD_BLOCKS=8334" "dtype=f64,SCHPF_UPD_BLOCKS=16384" "dtype=f64" 2>/dev/null | grep setting | python -c "$show" $cfg
 done
done | tee $O/ab_update_grid.txt
