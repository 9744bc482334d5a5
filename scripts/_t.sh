python -m pytest tests/test_gpu_optim.py tests/test_gpu_headline_config.py tests/test_gpu_data_parallel.py -x -q -m gpu 2>&1 | tail -3
python scripts/host_time.py 2>&1 | tail -7
( cd /tmp && python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-pmc 2>/dev/null | cut -c1-300 )
