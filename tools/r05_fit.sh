#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; O=gpurun_out/r05s; mkdir -p $O
SCHPF_VERBOSE=1 timeout 900 python tools/fitprof.py > $O/fitprof.txt 2>&1
grep -v "^\[schpf_hip\]     \|task ranges\|loss pass on" $O/fitprof.txt | head -80
