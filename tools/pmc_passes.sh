#!/bin/bash
# Wave-cycle breakdown of the sweep kernels (GPU box):  tools/pmc_passes.sh <tag> <bench args...>
# One rocprofv3 --pmc pass per counter group (never combined with traces), summaries into gpurun_out/.
tag=$1; shift
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for grp in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_VALU" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_BUSY_CYCLES" \
           "SQ_ACTIVE_INST_VMEM SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_SCA SQ_INSTS_SALU"; do
  name=$(echo $grp | tr ' ' '_' | cut -c1-40)
  out=$R/gpurun_out/pmc_${tag}_$name
  rm -rf $out
  timeout 600 rocprofv3 --pmc $grp -d $out -o pmc -- python $R/bench.py "$@" --no-cpu-baseline --no-converge --no-traffic > $out.log 2>&1
  db=$(find $out -name "*.db" | head -1)
  python $R/tools/rocpd_summary.py $db | grep -E "tile_sweep|counter" > $R/gpurun_out/pmc_${tag}_$name.txt
  rm -rf $out
done
cat $R/gpurun_out/pmc_${tag}_*.txt
