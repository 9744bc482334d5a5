set -x
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -4 > gpurun_out/pytest_gpu.txt
timeout 400 python bench.py > gpurun_out/r01_bench_n1.json 2> gpurun_out/r01_bench_n1.err
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof7 -o b -- python bench.py --steps 4 --warmup 2 --no-cpu-baseline > gpurun_out/r01_bench_n1_under_rocprof.json 2>/dev/null
f=$(find /tmp/prof7 -name 'b_kernel_stats.csv' | head -1); head -61 $f > gpurun_out/r01_bench_n1_kernel_stats.csv
timeout 300 python scripts/op_bench.py > gpurun_out/r01_op_bench.txt 2>&1
cat gpurun_out/pytest_gpu.txt; cat gpurun_out/r01_bench_n1.json | cut -c1-600
