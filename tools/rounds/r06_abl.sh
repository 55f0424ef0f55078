#!/bin/bash
# Round 6, third GPU call: (1) what the window barriers and stagings cost the C3 sweep -- development builds of the shipped
# kernel with the staging copies after a task's first window removed (1), the two barriers per (half) window removed
# (2), both (3); results are INVALID numerically, timing only (profiles/r06/ablate_sync.patch is the patch); (2) VALU /
# wave-cycle counters of the two update kernels (SCHPF_UPD=1 round 3, =2 round 6); (3) the tests added since the last call.
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; O=gpurun_out/r06; mkdir -p $O
for i in 1 2; do
  for a in 0 1 2 3; do
    SCHPF_LIB_PATH=$R/schpf_amd/libschpf_hip_dev_abl$a.so timeout 300 python tools/explore.py c3 "dtype=f64" "dtype=f32" 2>&1 | grep setting | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); print('ablate=$a', d['setting'], 'sweep', d['cell_ms'], 'iter', d['iter_ms'], 'loss', d['loss'])"
  done
done | tee $O/ablate_sync_c3.txt
for a in 0 1 2 3; do
  SCHPF_LIB_PATH=$R/schpf_amd/libschpf_hip_dev_abl$a.so timeout 300 python tools/explore.py c5-shard "dtype=f64" 2>&1 | grep setting | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); print('ablate=$a', d['setting'], 'sweep', d['cell_ms'], 'iter', d['iter_ms'], 'loss', d['loss'])"
done | tee $O/ablate_sync_c5shard.txt
cd /tmp && export TMPDIR=/tmp
for v in 1 2; do
  for grp in "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY" "SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_INSTS_SALU SQ_INSTS_LDS"; do
    out=$R/gpurun_out/pmc_upd$v; rm -rf $out
    timeout 600 rocprofv3 --pmc $grp -d $out -o pmc -- python $R/tools/explore.py c3 "dtype=f64,SCHPF_UPD=$v" > $out.log 2>&1
    echo "== SCHPF_UPD=$v"; python $R/tools/rocpd_summary.py $(find $out -name "*.db" | head -1) | grep -E "gamma_update|colsum|counter"
    rm -rf $out
  done
  out=$R/gpurun_out/stats_upd$v; rm -rf $out
  timeout 600 rocprofv3 --kernel-trace --stats -d $out -o st -- python $R/tools/explore.py c3 "dtype=f64,SCHPF_UPD=$v" > $out.log 2>&1
  echo "== SCHPF_UPD=$v kernel stats"; python $R/tools/rocpd_summary.py $(find $out -name "*.db" | head -1) | grep -E "gamma_update|colsum|dual|calls"
  rm -rf $out
done 2>&1 | tee $O/update_kernel_counters_c3_f64.txt
cd $R
timeout 900 python -m pytest tests/test_trajectory_gpu.py tests/test_multigpu.py -q -m gpu -k "c2_whole_fit or draws_c5 or bench_launches" > $O/pytest_new.log 2>&1; echo "pytest new rc $?"; tail -8 $O/pytest_new.log
