#!/bin/bash
# Round-end measurement on the GPU box: kernel stats and HBM-traffic counters for the bench command
# (one rocprofv3 pass per counter, never combined with traces), then plain bench runs.
#   tools/profile_round.sh <tag>          -> gpurun_out/<tag>_*
tag=${1:-r01}
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
BENCH="python $R/bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-converge --no-traffic"
rm -rf $O/${tag}_stats
timeout 900 rocprofv3 --kernel-trace --stats -d $O/${tag}_stats -o c3f64 -- $BENCH > $O/${tag}_bench_under_rocprof_c3_f64.json 2> $O/${tag}_stats.log
python $R/tools/rocpd_summary.py $(find $O/${tag}_stats -name "*.db" | head -1) > $O/${tag}_kernel_stats_c3_f64.txt 2>&1
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf $O/${tag}_pmc_$c
  timeout 900 rocprofv3 --pmc $c -d $O/${tag}_pmc_$c -o pmc -- $BENCH > /dev/null 2> $O/${tag}_pmc_$c.log
  python $R/tools/rocpd_summary.py $(find $O/${tag}_pmc_$c -name "*.db" | head -1) | grep -E "counter|sweep|gamma_update" > $O/${tag}_pmc_$c.txt 2>&1
  rm -rf $O/${tag}_pmc_$c
done
rm -rf $O/${tag}_stats
cd $R
python bench.py > $O/${tag}_bench_c3_f64.json 2> $O/${tag}_bench_c3_f64.err
python bench.py --dtype f32 --no-cpu-baseline > $O/${tag}_bench_c3_f32.json 2>> $O/${tag}_bench_c3_f64.err
python bench.py --config c2 --no-cpu-baseline --no-converge --no-traffic > $O/${tag}_bench_c2_f64.json 2>> $O/${tag}_bench_c3_f64.err
python bench.py --config c5-shard --no-cpu-baseline --no-converge --no-traffic --steps 30 --warmup 5 > $O/${tag}_bench_c5shard_f64.json 2>> $O/${tag}_bench_c3_f64.err
python bench.py --force-sharded --no-cpu-baseline --no-converge --no-traffic > $O/${tag}_bench_c3_f64_sharded1.json 2>> $O/${tag}_bench_c3_f64.err
python bench.py --config c4-shard --force-sharded --no-cpu-baseline --no-converge --no-traffic > $O/${tag}_bench_c4shard_f64_sharded1.json 2>> $O/${tag}_bench_c3_f64.err
python bench.py --config c4-shard4 --force-sharded --no-cpu-baseline --no-converge --no-traffic > $O/${tag}_bench_c4shard4_f64_sharded1.json 2>> $O/${tag}_bench_c3_f64.err
python bench.py --config c4-shard2 --force-sharded --no-cpu-baseline --no-converge --no-traffic > $O/${tag}_bench_c4shard2_f64_sharded1.json 2>> $O/${tag}_bench_c3_f64.err
python bench.py --config c5-shard --dtype f32 --no-cpu-baseline --no-converge --no-traffic --steps 30 --warmup 5 > $O/${tag}_bench_c5shard_f32.json 2>> $O/${tag}_bench_c3_f64.err
python bench.py --config c2 --dtype f32 --no-cpu-baseline --no-converge --no-traffic > $O/${tag}_bench_c2_f32.json 2>> $O/${tag}_bench_c3_f64.err
python bench.py --config c5 --no-cpu-baseline --no-converge --no-traffic --steps 20 --warmup 3 > $O/${tag}_bench_c5_whole_f64.json 2>> $O/${tag}_bench_c3_f64.err
for f in $O/${tag}_bench_*.json; do echo "== $f"; python - "$f" <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    r = d["roofline"]
    print(d["dtype"], "value %.1f" % d["value"], "ms %.4f" % d["ms_per_step"], "frac %.4f" % r["frac"], "launch_ms %.4f" % r["avg_launch_ms"], "upd %.4f" % r["gamma_updates_ms"], "loss_ms %.3f" % d["loss_eval_ms"], d.get("cpu_baseline"))
except Exception as e:
    print("unreadable:", e)
PY
done
cat $O/${tag}_kernel_stats_c3_f64.txt | cut -c1-150 | head -12
cat $O/${tag}_pmc_*.txt
