cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_engine_gpu.py -x -q -m gpu -k "duplicate" 2>&1 | tail -3
SCHPF_VERBOSE=1 python tools/explore.py c5-shard "dtype=f64" 2>&1 | grep -E "balance|balanced|setting" | cut -c1-200
