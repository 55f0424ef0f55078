#!/bin/bash
# Round 5, first GPU call: the whole -m gpu suite (new C4-at-size tests included), then the driver-style bench line.
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; O=gpurun_out/r05; mkdir -p $O
python -c "import __graft_entry__ as g; g.build(); g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc $?"
timeout 2400 python -m pytest tests/ -x -q -m gpu --durations=25 > $O/pytest_gpu.log 2>&1; echo "pytest rc $?"
tail -40 $O/pytest_gpu.log
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver_style.json 2> $O/bench_driver_style.err; echo "bench rc $?"
tail -c 3000 $O/bench_driver_style.json
