#!/bin/bash
# Round 5, first GPU call: the whole -m gpu suite (new C4-at-size tests included), the driver-style bench line, and the
# micro-benchmarks whose tables the round-4 verdict asked for (grid barrier vs graph node, single-pass hand-over).
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; O=gpurun_out/r05; mkdir -p $O
python -c "import __graft_entry__ as g; g.build(); g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc $?"
timeout 2400 python -m pytest tests/ -x -q -m gpu --durations=25 > $O/pytest_gpu.log 2>&1; echo "pytest rc $?"
tail -40 $O/pytest_gpu.log
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver_style.json 2> $O/bench_driver_style.err; echo "bench rc $?"
tail -c 1500 $O/bench_driver_style.json
for m in grid_barrier handover_bench; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 tools/micro/$m.hip -o /tmp/$m && timeout 300 /tmp/$m > $O/micro_$m.txt 2>&1; echo "$m rc $?"
  cat $O/micro_$m.txt
done
