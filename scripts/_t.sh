python -m pytest tests/test_gpu_conv.py tests/test_gpu_detector.py -x -q -m gpu --tb=short -k "class_row or backward_forms" 2>&1 | tail -4
cd /tmp; export TMPDIR=/tmp; rm -rf /tmp/prof7
FI_WGRAD_SIDE_PIXELS=0 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof7 -o b -- python $GRAFT_REPO_ROOT/bench.py --steps 4 --warmup 2 --profile-steps 0 --no-cpu-baseline --no-pmc > /dev/null 2>&1
f=$(find /tmp/prof7 -name 'b_kernel_stats.csv' | head -1); grep -E "class_row|Name" $f | cut -c1-260
cd $GRAFT_REPO_ROOT; bash scripts/ab_env.sh "FI_X=1" "FI_X=2" 2>&1 | grep -v amdgpu
