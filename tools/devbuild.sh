#!/bin/bash
# Development build: only the K = 20 sweep instantiations (seconds instead of minutes per file).
#   tools/devbuild.sh            -> schpf_amd/libschpf_hip_dev.so  (use with SCHPF_LIB_PATH=...)

set -e
cd "$(dirname "$0")/../schpf_amd/csrc"
FLAGS="-O3 -std=c++17 --offload-arch=gfx950 -fPIC -Wall -Wno-unused-function"
mkdir -p /tmp/schpf_dev
for f in sweep_f64 sweep_f32; do
  /opt/rocm/bin/hipcc $FLAGS -DSCHPF_DEV_FAST $DEVFLAGS -c $f.hip -o /tmp/schpf_dev/$f.o &
done
for f in kernels capi plan_device; do
  /opt/rocm/bin/hipcc $FLAGS -c $f.hip -o /tmp/schpf_dev/$f.o &
done
/opt/rocm/bin/hipcc $FLAGS -x hip -c plan.cpp -o /tmp/schpf_dev/plan.o &
wait
g++ -shared -fPIC -o ../libschpf_hip_dev.so /tmp/schpf_dev/*.o
echo built ../libschpf_hip_dev.so
