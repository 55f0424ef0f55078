#!/bin/bash
# K = 50 (C5 share, 125k x 25k, 2 %, f64) under rocprofv3 on the GPU box: kernel stats, HBM traffic, SQ counters of the
# shipped sweep kernel.   tools/profile_k50.sh <tag> [dtype]   -> gpurun_out/<tag>_*c5shard_<dtype>.txt
tag=${1:-r04}; dt=${2:-f64}
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
BENCH="python $R/bench.py --config c5-shard --dtype $dt --steps 20 --warmup 3 --no-cpu-baseline --no-converge --no-traffic"
rm -rf $O/${tag}_k50stats
timeout 600 rocprofv3 --kernel-trace --stats -d $O/${tag}_k50stats -o k50 -- $BENCH > $O/${tag}_bench_under_rocprof_c5shard_$dt.json 2> $O/${tag}_k50stats.log
python $R/tools/rocpd_summary.py $(find $O/${tag}_k50stats -name "*.db" | head -1) > $O/${tag}_rocprofv3_kernel_stats_c5shard_$dt.txt 2>&1
rm -rf $O/${tag}_k50stats
: > $O/${tag}_rocprofv3_pmc_traffic_c5shard_$dt.txt
for c in FETCH_SIZE WRITE_SIZE "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum"; do
  out=$O/${tag}_k50pmc
  rm -rf $out
  timeout 600 rocprofv3 --pmc $c -d $out -o pmc -- $BENCH > /dev/null 2> $out.log
  python $R/tools/rocpd_summary.py $(find $out -name "*.db" | head -1) | grep -E "counter|tile_sweep|gamma_update" >> $O/${tag}_rocprofv3_pmc_traffic_c5shard_$dt.txt 2>&1
  rm -rf $out
done
: > $O/${tag}_rocprofv3_pmc_sq_counters_c5shard_$dt.txt
for grp in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_VALU" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_BUSY_CYCLES" \
           "SQ_ACTIVE_INST_VMEM SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_SCA SQ_INSTS_SALU"; do
  out=$O/${tag}_k50pmc
  rm -rf $out
  timeout 600 rocprofv3 --pmc $grp -d $out -o pmc -- $BENCH > /dev/null 2> $out.log
  python $R/tools/rocpd_summary.py $(find $out -name "*.db" | head -1) | grep -E "counter|tile_sweep" >> $O/${tag}_rocprofv3_pmc_sq_counters_c5shard_$dt.txt 2>&1
  rm -rf $out
done
cd $R
python bench.py --config c5-shard --dtype $dt --steps 60 --warmup 10 --no-cpu-baseline --no-converge --no-traffic > $O/${tag}_bench_c5shard_$dt.json 2> $O/${tag}_bench_c5shard_$dt.err
cat $O/${tag}_rocprofv3_kernel_stats_c5shard_$dt.txt | cut -c1-160 | head -14
cat $O/${tag}_rocprofv3_pmc_traffic_c5shard_$dt.txt $O/${tag}_rocprofv3_pmc_sq_counters_c5shard_$dt.txt
tail -1 $O/${tag}_bench_c5shard_$dt.json | cut -c1-600
