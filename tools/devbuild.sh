#!/bin/bash
# Development build: only the K = 20 / K = 50 sweep instantiations (seconds instead of minutes per file).
#   tools/devbuild.sh [tag]      -> schpf_amd/libschpf_hip_dev[_tag].so  (use with SCHPF_LIB_PATH=...)
set -e
tag=${1:+_$1}
cd "$(dirname "$0")/../schpf_amd/csrc"
FLAGS="-O3 -std=c++17 --offload-arch=gfx950 -fPIC -Wall -Wno-unused-function"
D=/tmp/schpf_dev$tag
mkdir -p $D
for f in sweep_f64 sweep_f32; do
  /opt/rocm/bin/hipcc $FLAGS -DSCHPF_DEV_FAST $DEVFLAGS -c $f.hip -o $D/$f.o &
done
for f in kernels capi plan_device; do
  if [ ! -f /tmp/schpf_dev_common/$f.o ] || [ $f.hip -nt /tmp/schpf_dev_common/$f.o ] || [ plan.h -nt /tmp/schpf_dev_common/$f.o ] || [ kernels.h -nt /tmp/schpf_dev_common/$f.o ]; then
    mkdir -p /tmp/schpf_dev_common
    /opt/rocm/bin/hipcc $FLAGS -c $f.hip -o /tmp/schpf_dev_common/$f.o &
  fi
done
if [ ! -f /tmp/schpf_dev_common/plan.o ] || [ plan.cpp -nt /tmp/schpf_dev_common/plan.o ] || [ plan.h -nt /tmp/schpf_dev_common/plan.o ]; then
  mkdir -p /tmp/schpf_dev_common
  /opt/rocm/bin/hipcc $FLAGS -x hip -c plan.cpp -o /tmp/schpf_dev_common/plan.o &
fi
wait
g++ -shared -fPIC -o ../libschpf_hip_dev$tag.so $D/sweep_f64.o $D/sweep_f32.o /tmp/schpf_dev_common/*.o
echo built ../libschpf_hip_dev$tag.so
