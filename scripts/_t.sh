python -m pytest tests/test_gpu_detector.py tests/test_gpu_headline_config.py -x -q -m gpu --tb=short 2>&1 | tail -8
bash scripts/ab_env.sh "FI_X=1" "FI_X=2" 2>&1 | grep -v amdgpu
