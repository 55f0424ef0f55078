#!/bin/bash
# Round 6: the whole -m gpu suite four more times with the native backtrace handler on (one run in five died of a
# segmentation fault early in the round and never again: profiles/r06/README.md, last section).
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; O=gpurun_out/r06; mkdir -p $O
python -c "import __graft_entry__ as g; g.build(); g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc $?"
for i in 3 4 5 6; do
  SCHPF_BACKTRACE=1 timeout 2400 python -m pytest tests/ -x -q -m gpu > $O/pytest_gpu_bt$i.log 2>&1; echo "pytest $i rc $?"
  grep -n "schpf_hip\] fatal" -A30 $O/pytest_gpu_bt$i.log | head -50; tail -2 $O/pytest_gpu_bt$i.log | cut -c1-200
done
