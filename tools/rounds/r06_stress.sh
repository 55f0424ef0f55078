#!/bin/bash
# Round 6: the segmentation fault of the first full-suite run did not come back in the second; the capture path under
# volume (tools/stress_graph.py) and the whole suite twice more, all with the native backtrace handler on.
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; O=gpurun_out/r06; mkdir -p $O
python -c "import __graft_entry__ as g; g.build(); g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc $?"
SCHPF_BACKTRACE=1 timeout 900 python tools/stress_graph.py 24 > $O/stress_graph.log 2>&1; echo "stress rc $?"; tail -4 $O/stress_graph.log | cut -c1-200
for i in 1 2; do
  SCHPF_BACKTRACE=1 timeout 2400 python -m pytest tests/ -x -q -m gpu > $O/pytest_gpu_bt$i.log 2>&1; echo "pytest $i rc $?"
  grep -n "schpf_hip\] fatal" -A30 $O/pytest_gpu_bt$i.log | head -50; tail -3 $O/pytest_gpu_bt$i.log | cut -c1-200
done
