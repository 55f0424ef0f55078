#!/bin/bash
# A/B of two library builds on the same box (boxes differ by +-5 %, so never compare across gpurun
# calls): keep the reference build as schpf_amd/libschpf_hip_base.so, build the variant, run this.
R=$GRAFT_REPO_ROOT
for i in 1 2; do
  for lib in libschpf_hip_base.so libschpf_hip.so; do
    SCHPF_LIB_PATH=$R/schpf_amd/$lib python $R/tools/explore.py c3 "dtype=f64" "dtype=f32" 2>&1 | grep setting | cut -c1-110 | sed "s/^/$lib /"
  done
done
