#!/bin/bash
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
bash $R/tools/pmc_passes.sh c5 --config c5-shard --steps 10 --warmup 2 > /dev/null 2>&1
for c in FETCH_SIZE "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" "TCP_TCC_READ_REQ_sum TCC_EA_RDREQ_sum TCC_EA_RDREQ_32B_sum"; do
  name=$(echo $c | tr ' ' '_' | cut -c1-30)
  out=$R/gpurun_out/pmc_c5_$name
  rm -rf $out
  timeout 600 rocprofv3 --pmc $c -d $out -o pmc -- python $R/bench.py --config c5-shard --steps 10 --warmup 2 --no-cpu-baseline --no-converge --no-traffic > $out.log 2>&1
  python $R/tools/rocpd_summary.py $(find $out -name "*.db" | head -1) | grep -E "counter|tile_sweep" > $R/gpurun_out/pmc_c5_$name.txt 2>&1
  rm -rf $out
done
cat $R/gpurun_out/pmc_c5_*.txt
