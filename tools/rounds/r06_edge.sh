#!/bin/bash
# Round 6: the two edge-case tests added late (a matrix without stored entries, K = 256) on the clean rebuild of the library.
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; O=gpurun_out/r06; mkdir -p $O
python -c "import __graft_entry__ as g; g.build(); g.smoke()" 2>&1 | tail -1
SCHPF_BACKTRACE=1 timeout 900 python -m pytest tests/test_engine_gpu.py -x -q -m gpu -k "without_stored or largest_supported or engine_argument or empty_rows" > $O/pytest_edge.log 2>&1; echo "pytest edge rc $?"; tail -25 $O/pytest_edge.log | cut -c1-220
