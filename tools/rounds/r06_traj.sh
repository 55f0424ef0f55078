#!/bin/bash
# Round 6, first GPU call: the trajectory-parity tests (tests/test_trajectory_gpu.py) with their drift records, and the
# driver-style bench line of the tree as it stands (same box: the baseline of this round's A/Bs).
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; O=gpurun_out/r06; mkdir -p $O
nproc; free -g | head -2
python -c "import __graft_entry__ as g; g.build(); g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc $?"
timeout 1500 python -m pytest tests/test_trajectory_gpu.py -q -m gpu --durations=10 > $O/pytest_traj.log 2>&1; echo "pytest rc $?"
tail -60 $O/pytest_traj.log
ls gpurun_out/trajectory
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 --no-converge > $O/bench_driver_style_first.json 2> $O/bench_first.err; echo "bench rc $?"
tail -c 600 $O/bench_driver_style_first.json
