#!/bin/bash
# One rocprofv3 --pmc pass (GPU box):  tools/pmc_one.sh <tag> "<counters>" <bench args...>
tag=$1; ctr=$2; shift 2
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
out=$R/gpurun_out/pmc1_$tag
rm -rf $out
timeout 600 rocprofv3 --pmc $ctr -d $out -o pmc -- python $R/bench.py "$@" --no-cpu-baseline --no-converge --no-traffic > $out.log 2>&1
python $R/tools/rocpd_summary.py $(find $out -name "*.db" | head -1) | grep -E "tile_sweep_dual|counter" > $R/gpurun_out/pmc1_$tag.txt
rm -rf $out
cat $R/gpurun_out/pmc1_$tag.txt
