#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; O=gpurun_out/r05l; mkdir -p $O
for cfg in c3 c5-shard c2 c4-shard; do
  for dt in f64 f32; do
    SCHPF_VERBOSE=1 timeout 600 python tools/loss_time.py $cfg "dtype=$dt" "dtype=$dt,SCHPF_LOSS_SIDE=0" "dtype=$dt,SCHPF_LOSS_SIDE=1" "dtype=$dt,SCHPF_LOSS_SPLIT=0,SCHPF_LOSS_SIDE=0" "dtype=$dt,SCHPF_LOSS_SPLIT=0,SCHPF_LOSS_SIDE=1" 2> $O/loss_$cfg_$dt.err | sed "s/^/$cfg /"
    grep "loss pass on" $O/loss_$cfg_$dt.err | sed "s/^/$cfg $dt /"
  done
done 2>&1 | tee $O/loss_model_calibration.txt
