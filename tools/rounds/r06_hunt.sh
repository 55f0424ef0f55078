#!/bin/bash
# Round 6: the whole -m gpu suite three more times per call with the native backtrace handler on (the one segmentation
# fault of the round's first full run never came back: profiles/r06/README.md, last section).
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; O=gpurun_out/r06; mkdir -p $O
tag=${1:-a}
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
for i in 1 2 3; do
  SCHPF_BACKTRACE=1 timeout 1500 python -m pytest tests/ -x -q -m gpu > $O/pytest_hunt_$tag$i.log 2>&1; echo "pytest $tag$i rc $?"
  grep -n "schpf_hip\] fatal" -A40 $O/pytest_hunt_$tag$i.log | head -60; tail -2 $O/pytest_hunt_$tag$i.log | cut -c1-200
done
