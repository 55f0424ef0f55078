#!/bin/bash
# A/B on ONE box: balanced windows (plan.h) against windows cut by index, tools/explore.py lines.
R=$GRAFT_REPO_ROOT
cd $R
for i in 1 2; do
  SCHPF_VERBOSE=${VERBOSE:-0} python tools/explore.py c5-shard "dtype=f64" "dtype=f64,SCHPF_BALANCE=0" "dtype=f32" "dtype=f32,SCHPF_BALANCE=0" 2>&1 | grep -E "setting|balanced|ELL fill|error|Error" | cut -c1-330
done
[ -n "$SKIP_C3" ] || SCHPF_VERBOSE=${VERBOSE:-0} python tools/explore.py c3 "dtype=f64" "dtype=f64,SCHPF_BALANCE=1" "dtype=f32" "dtype=f32,SCHPF_BALANCE=1" 2>&1 | grep -E "setting|balanced|ELL fill|error|Error" | cut -c1-330
