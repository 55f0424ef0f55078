#!/bin/bash
# One bench configuration under rocprofv3 on the GPU box: kernel stats, HBM-side traffic (one counter per pass) and the
# SQ counters of the sweep kernel (four groups, one pass each; never combined with traces).
#   tools/profile_counters.sh <tag> <config> <dtype> [steps]   -> gpurun_out/<tag>/rocprofv3_*_<config>_<dtype>.txt
tag=${1:-r05}; cfg=${2:-c3}; dt=${3:-f64}; steps=${4:-20}
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/$tag
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
name=$(echo $cfg | tr -d '-')_$dt
BENCH="python $R/bench.py --config $cfg --dtype $dt --steps $steps --warmup 3 --no-cpu-baseline --no-converge --no-traffic"
w=$O/_work_$name
rm -rf $w
timeout 900 rocprofv3 --kernel-trace --stats -d $w -o st -- $BENCH > $O/bench_under_rocprof_$name.json 2> $w.log
python $R/tools/rocpd_summary.py $(find $w -name "*.db" | head -1) > $O/rocprofv3_kernel_stats_$name.txt 2>&1
rm -rf $w
: > $O/rocprofv3_pmc_traffic_$name.txt
for c in FETCH_SIZE WRITE_SIZE "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum"; do
  rm -rf $w
  timeout 900 rocprofv3 --pmc $c -d $w -o pmc -- $BENCH > /dev/null 2> $w.log
  python $R/tools/rocpd_summary.py $(find $w -name "*.db" | head -1) | grep -E "counter|tile_sweep|gamma_update" >> $O/rocprofv3_pmc_traffic_$name.txt 2>&1
  rm -rf $w
done
: > $O/rocprofv3_pmc_sq_counters_$name.txt
for grp in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_VALU" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_BUSY_CYCLES" \
           "SQ_ACTIVE_INST_VMEM SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_SCA SQ_INSTS_SALU"; do
  rm -rf $w
  timeout 900 rocprofv3 --pmc $grp -d $w -o pmc -- $BENCH > /dev/null 2> $w.log
  python $R/tools/rocpd_summary.py $(find $w -name "*.db" | head -1) | grep -E "counter|tile_sweep" >> $O/rocprofv3_pmc_sq_counters_$name.txt 2>&1
  rm -rf $w
done
rm -f $w.log
head -8 $O/rocprofv3_kernel_stats_$name.txt | cut -c1-170
cat $O/rocprofv3_pmc_traffic_$name.txt $O/rocprofv3_pmc_sq_counters_$name.txt
